// Microbenchmark: fp32-accurate implicit-GEMM conv on the bf16 MFMA pipe with PRE-SPLIT operands (a = a0 + a1 + a2, bf16 each).
//   Y[b][m][n] = sum_tap sum_c W[tap][m][c] * X[b][c][n + tap - pad]
// Both operands live in HBM as 16-byte chunks of 8 consecutive channels of one plane:
//   Wp [tap][C/8][3][M ][8 bf16]      Xp [b][C/8][3][Tp][8 bf16]   (Tp = padded time axis with a zero halo)
// so a K-step (16 channels) of a 128-row tile is 6 "kinds" (plane, k-half) x 128 rows x 16 B = 12 KiB, every kind a contiguous
// 2 KiB run in HBM *and* in LDS: the whole tile is moved by global_load_lds (no VGPR staging), and a ds_read_b128 of
// (kind, row = lane&31) is bank-conflict-free.  3-stage pipeline, one barrier per K-step, 24 MFMA 32x32x16 per wave per K-step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int BM = 128, BN = 128, TILE = 6 * 128 * 16;

// WM: waves along M (2 -> 128x128 block, 4 waves; 4 -> 256x128 block, 8 waves).  One LDS-DMA load after every group of 4 MFMAs.
// RD: 0 = all fragment reads up front; 1 = the first term's operands first, the rest behind the first MFMA group
template <int WM, int RD, int ABL = 0>
__global__ __launch_bounds__(WM * 128) void gemm_x3(const uint4* __restrict__ Wp, const uint4* __restrict__ Xp, float* __restrict__ Y, int M,
                                                    int C8, int taps, int Tp, int T, int xoff) {
    constexpr int BMk = 64 * WM, ATILE = 6 * BMk * 16, BTILE = 6 * 128 * 16, STAGE = ATILE + BTILE, ARH = BMk / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int nwg = gridDim.x * gridDim.y * gridDim.z, lin = (bz * gridDim.y + by) * gridDim.x + bx;
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = v % gridDim.x; by = (v / gridDim.x) % gridDim.y; bz = v / (gridDim.x * gridDim.y);
    }
    const int m0 = bx * BMk, n0 = by * BN, b = bz;
    const int c16n = C8 / 2, nks = taps * c16n;
    // DMA ownership: WM=2: waves 0,1 -> A (6 each), 2,3 -> B (6 each);  WM=4: waves 0..5 -> A (4 each), 6,7 -> B (6 each)
    const int nA = WM == 2 ? 2 : 6;
    const int operand = wave >= nA;
    const int cnt = operand ? 6 : (WM == 2 ? 6 : 4);
    const int jbase = operand ? (wave - nA) * 6 : wave * cnt;
    const uint4* gbase = (operand ? Xp + (size_t)b * C8 * 3 * Tp + n0 + xoff : Wp + m0) + lane;
    const long long rowlen = operand ? Tp : M;
    const long long tapstride = operand ? 1 : (long long)C8 * 3 * M;
    int ks_n = 0, tap_n = 0, c16_n = 0;
    auto issue_one = [&](int i, int stage) {
        if (i >= cnt) return;
        const int j = jbase + i;
        const int kind = operand ? j >> 1 : j / ARH, rh = operand ? j & 1 : j % ARH, p = kind >> 1, h = kind & 1;
        const uint4* g = gbase + tap_n * tapstride + ((long long)(2 * c16_n + h) * 3 + p) * rowlen + rh * 64;
        unsigned char* l = smem + stage * STAGE + (operand ? ATILE + kind * 2048 : kind * (BMk * 16)) + rh * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    };
    auto advance = [&]() {
        if (ks_n + 1 < nks) { ++ks_n; if (++c16_n == c16n) { c16_n = 0; ++tap_n; } }
    };
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_one(i, 0);
    advance();
    for (int ks = 0; ks < nks; ++ks) {
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const int nst = (ks + 1) & 1;
        const unsigned char* As = smem + (ks & 1) * STAGE + lhi * (BMk * 16);
        const unsigned char* Bs = smem + (ks & 1) * STAGE + ATILE + lhi * 2048;
        bf8 a[2][3], bb[2][3];
        auto lda = [&](int i, int p) { a[i][p] = *reinterpret_cast<const bf8*>(As + p * (2 * BMk * 16) + (wm0 + i * 32 + l31) * 16); };
        auto ldb = [&](int j, int p) { bb[j][p] = *reinterpret_cast<const bf8*>(Bs + p * 4096 + (wn0 + j * 32 + l31) * 16); };
        if (RD == 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p) { lda(0, p); lda(1, p); ldb(0, p); ldb(1, p); }
        } else {
            lda(0, 2); lda(1, 2); ldb(0, 0); ldb(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            lda(0, 1); lda(1, 1); ldb(0, 1); ldb(1, 1);
            lda(0, 0); lda(1, 0); ldb(0, 2); ldb(1, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) issue_one(t, nst);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* yb = Y + (long long)b * M * T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                if (n < T) yb[(long long)row * T + n] = acc[i][j][r];
            }
}

// Strength-reduced K loop (1-tap): all DMA source pointers are running per-lane pointers bumped by a constant per K-step, the
// loop is unrolled by two so that every LDS address (stage parity) is an immediate offset: ~30 instead of ~170 integer
// instructions per 24 MFMAs.
__global__ __launch_bounds__(256) void gemm_x3_sr(const uint4* __restrict__ Wp, const uint4* __restrict__ Xp, float* __restrict__ Y, int M,
                                                  int C8, int taps, int Tp, int T, int xoff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int nwg = gridDim.x * gridDim.y * gridDim.z, lin = (bz * gridDim.y + by) * gridDim.x + bx;
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = v % gridDim.x; by = (v / gridDim.x) % gridDim.y; bz = v / (gridDim.x * gridDim.y);
    }
    const int m0 = bx * BM, n0 = by * BN, b = bz;
    const int nks = C8 / 2;                            // taps == 1
    const int operand = wave >> 1;
    const long long rowlen = operand ? Tp : M;
    // six running source pointers (per lane), one per piece this wave owns; each K-step advances them by 2*3*rowlen chunks
    const uint4* src[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = (wave & 1) * 6 + i, kind = j >> 1, p = kind >> 1, h = kind & 1, rh = j & 1;
        src[i] = (operand ? Xp + (size_t)b * C8 * 3 * Tp + n0 + xoff : Wp + m0) + lane + ((long long)h * 3 + p) * rowlen + rh * 64;
    }
    const long long kstride = 6 * rowlen;
    const int lds_piece0 = operand * TILE + ((wave & 1) * 6) * 1024;     // pieces j -> kind*2048 + rh*1024 == j*1024
    auto issue_one = [&](int i, int stage) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                         (__attribute__((address_space(3))) void*)(smem + stage * 2 * TILE + lds_piece0 + i * 1024), 16, 0, 0);
        src[i] += kstride;
    };
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_one(i, 0);
    const unsigned char* a_base = smem + lhi * 2048 + (wm0 + l31) * 16;
    const unsigned char* b_base = smem + TILE + lhi * 2048 + (wn0 + l31) * 16;
    constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
    auto kstep = [&](int stage, bool fetch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf8 a[2][3], bb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][p] = *reinterpret_cast<const bf8*>(a_base + stage * 2 * TILE + p * 4096 + i * 512);
                bb[i][p] = *reinterpret_cast<const bf8*>(b_base + stage * 2 * TILE + p * 4096 + i * 512);
            }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (fetch) issue_one(t, stage ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int ks = 0;
    for (; ks + 2 < nks; ks += 2) {
        kstep(0, true);
        kstep(1, true);
    }
    for (; ks < nks; ++ks) kstep(ks & 1, ks + 1 < nks);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* yb = Y + (long long)b * M * T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                if (n < T) yb[(long long)row * T + n] = acc[i][j][r];
            }
}

// 128 x 64 block (wave tile 64 x 32): twice the workgroups of the 128 x 128 form at the same problem size, 18 DMA pieces per K-step
template <int DUMMY>
__global__ __launch_bounds__(256) void gemm_x3_n64(const uint4* __restrict__ Wp, const uint4* __restrict__ Xp, float* __restrict__ Y, int M,
                                                   int C8, int taps, int Tp, int T, int xoff) {
    constexpr int ATILE = 6 * 128 * 16, BTILE = 6 * 64 * 16, STAGE = ATILE + BTILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int nwg = gridDim.x * gridDim.y * gridDim.z, lin = (bz * gridDim.y + by) * gridDim.x + bx;
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        bx = v % gridDim.x; by = (v / gridDim.x) % gridDim.y; bz = v / (gridDim.x * gridDim.y);
    }
    const int m0 = bx * 128, n0 = by * 64, b = bz;
    const int c16n = C8 / 2, nks = taps * c16n;
    const uint4* xbase = Xp + (size_t)b * C8 * 3 * Tp + n0 + xoff + lane;
    const uint4* wbase = Wp + m0 + lane;
    const long long wtap = (long long)C8 * 3 * M;
    int ks_n = 0, tap_n = 0, c16_n = 0;
    // 18 pieces: W 12 (kind*2 + rh), X 6 (kind); wave w: W pieces 3w..3w+2, X pieces: waves 0,1 take 2 each (kinds 2w, 2w+1), waves 2,3 one each (kinds 2+w)
    auto issue_one = [&](int i, int stage) {
        if (i < 3) {
            const int j = wave * 3 + i, kind = j >> 1, p = kind >> 1, h = kind & 1, rh = j & 1;
            const uint4* g = wbase + tap_n * wtap + ((long long)(2 * c16_n + h) * 3 + p) * M + rh * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + kind * 2048 + rh * 1024), 16, 0, 0);
        } else {
            const int q = i - 3;
            int kind = -1;
            if (wave < 2) kind = 2 * wave + q; else if (q == 0) kind = 2 + wave;
            if (kind >= 0 && q < 2) {
                const int p = kind >> 1, h = kind & 1;
                const uint4* g = xbase + tap_n + ((long long)(2 * c16_n + h) * 3 + p) * Tp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + ATILE + kind * 1024), 16, 0, 0);
            }
        }
    };
    auto advance = [&]() { if (ks_n + 1 < nks) { ++ks_n; if (++c16_n == c16n) { c16_n = 0; ++tap_n; } } };
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 32;
    f16v acc[2];
    bf8 a[2][3], bb[3];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_one(i, 0);
    advance();
    for (int ks = 0; ks < nks; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nst = (ks + 1) & 1;
        const unsigned char* As = smem + (ks & 1) * STAGE + lhi * 2048;
        const unsigned char* Bs = smem + (ks & 1) * STAGE + ATILE + lhi * 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[0][p] = *reinterpret_cast<const bf8*>(As + p * 4096 + (wm0 + l31) * 16);
            a[1][p] = *reinterpret_cast<const bf8*>(As + p * 4096 + (wm0 + 32 + l31) * 16);
            bb[p] = *reinterpret_cast<const bf8*>(Bs + p * 2048 + (wn0 + l31) * 16);
        }
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], bb[TB[t]], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (t < 5) issue_one(t, nst);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* yb = Y + (long long)b * M * T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + l31;
            if (n < T) yb[(long long)row * T + n] = acc[i][r];
        }
}

static unsigned short h_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float v, unsigned short* p) {
    p[0] = h_bf16(v); const float r1 = v - h_f(p[0]);
    p[1] = h_bf16(r1); const float r2 = r1 - h_f(p[1]);
    p[2] = h_bf16(r2);
}

template <int WM, int RD, int ABL = 0>
void run(const uint4* Wp, const uint4* Xp, float* Y, int M, int C, int taps, int T, int Tp, int B, int nstream, const std::vector<float>& hw,
         const std::vector<float>& hx, const char* name) {
    const int pad = taps / 2, halo = 1;
    dim3 grid(M / (64 * WM), (T + BN - 1) / BN, B);
    const size_t lds = (size_t)2 * (6 * 64 * WM * 16 + 6 * 128 * 16);
    (void)hipFuncSetAttribute((const void*)gemm_x3<WM, RD, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm_x3<WM, RD, ABL>), grid, dim3(WM * 128), lds, 0, Wp, Xp, Y, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    std::vector<float> hy((size_t)M * T);
    (void)hipMemcpy(hy.data(), Y + (size_t)(B - 1) * M * T, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0;
    const float* xb = hx.data() + (size_t)(B - 1) * C * T;
    for (int m = 0; m < M; m += 37)
        for (int n = 0; n < T; n += 53) {
            double ref = 0;
            for (int tap = 0; tap < taps; ++tap) {
                const int t = n + tap - pad;
                if (t < 0 || t >= T) continue;
                for (int c = 0; c < C; ++c) ref += (double)hw[((size_t)tap * M + m) * C + c] * (double)xb[(size_t)c * T + t];
            }
            maxerr = fmax(maxerr, fabs(ref - hy[(size_t)m * T + n]));
            scale = fmax(scale, fabs(ref));
        }
    hipStream_t st[2]; (void)hipStreamCreate(&st[0]); (void)hipStreamCreate(&st[1]);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
        for (int k = 0; k < nstream; ++k)
            hipLaunchKernelGGL((gemm_x3<WM, RD, ABL>), grid, dim3(WM * 128), lds, nstream > 1 ? st[k] : 0, Wp, Xp + (size_t)k * B * (C / 8) * 3 * Tp,
                               Y + (size_t)k * B * M * T, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * M * C * taps * (double)T * B * reps * nstream;
    printf("%-28s taps %d B %2d x%d stream  max err %.3e (rel %.2e)  %7.1f us/launch-set  %6.1f TFLOP/s fp32-equivalent\n", name, taps, B, nstream,
           maxerr, maxerr / scale, ms / reps * 1e3, fl / (ms * 1e-3) / 1e12);
}

void run64(const uint4* Wp, const uint4* Xp, float* Y, int M, int C, int taps, int T, int Tp, int B, int nstream, const std::vector<float>& hw,
           const std::vector<float>& hx, const char* name) {
    const int pad = taps / 2, halo = 1;
    dim3 grid(M / 128, (T + 63) / 64, B);
    const size_t lds = (size_t)2 * (6 * 128 * 16 + 6 * 64 * 16);
    hipLaunchKernelGGL((gemm_x3_n64<0>), grid, dim3(256), lds, 0, Wp, Xp, Y, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    std::vector<float> hy((size_t)M * T);
    (void)hipMemcpy(hy.data(), Y + (size_t)(B - 1) * M * T, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0;
    const float* xb = hx.data() + (size_t)(B - 1) * C * T;
    for (int m = 0; m < M; m += 37)
        for (int n = 0; n < T; n += 53) {
            double ref = 0;
            for (int tap = 0; tap < taps; ++tap) {
                const int t = n + tap - pad;
                if (t < 0 || t >= T) continue;
                for (int c = 0; c < C; ++c) ref += (double)hw[((size_t)tap * M + m) * C + c] * (double)xb[(size_t)c * T + t];
            }
            maxerr = fmax(maxerr, fabs(ref - hy[(size_t)m * T + n]));
            scale = fmax(scale, fabs(ref));
        }
    hipStream_t st[2]; (void)hipStreamCreate(&st[0]); (void)hipStreamCreate(&st[1]);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
        for (int k = 0; k < nstream; ++k)
            hipLaunchKernelGGL((gemm_x3_n64<0>), grid, dim3(256), lds, nstream > 1 ? st[k] : 0, Wp, Xp + (size_t)k * B * (C / 8) * 3 * Tp,
                               Y + (size_t)k * B * M * T, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * M * C * taps * (double)T * B * reps * nstream;
    printf("%-28s taps %d B %2d x%d stream  max err %.3e (rel %.2e)  %7.1f us/launch-set  %6.1f TFLOP/s fp32-equivalent\n", name, taps, B, nstream,
           maxerr, maxerr / scale, ms / reps * 1e3, fl / (ms * 1e-3) / 1e12);
}

void runsr(const uint4* Wp, const uint4* Xp, float* Y, int M, int C, int T, int Tp, int B, int nstream, const std::vector<float>& hw,
           const std::vector<float>& hx, const char* name) {
    const int taps = 1, pad = 0, halo = 1;
    dim3 grid(M / 128, (T + 127) / 128, B);
    const size_t lds = (size_t)2 * 2 * TILE;
    hipLaunchKernelGGL(gemm_x3_sr, grid, dim3(256), lds, 0, Wp, Xp, Y, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    std::vector<float> hy((size_t)M * T);
    (void)hipMemcpy(hy.data(), Y + (size_t)(B - 1) * M * T, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0;
    const float* xb = hx.data() + (size_t)(B - 1) * C * T;
    for (int m = 0; m < M; m += 37)
        for (int n = 0; n < T; n += 53) {
            double ref = 0;
            for (int c = 0; c < C; ++c) ref += (double)hw[(size_t)m * C + c] * (double)xb[(size_t)c * T + n];
            maxerr = fmax(maxerr, fabs(ref - hy[(size_t)m * T + n]));
            scale = fmax(scale, fabs(ref));
        }
    hipStream_t st[2]; (void)hipStreamCreate(&st[0]); (void)hipStreamCreate(&st[1]);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
        for (int k = 0; k < nstream; ++k)
            hipLaunchKernelGGL(gemm_x3_sr, grid, dim3(256), lds, nstream > 1 ? st[k] : 0, Wp, Xp + (size_t)k * B * (C / 8) * 3 * Tp,
                               Y + (size_t)k * B * M * T, M, C / 8, taps, Tp, T, halo - pad);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * M * C * taps * (double)T * B * reps * nstream;
    printf("%-28s taps %d B %2d x%d stream  max err %.3e (rel %.2e)  %7.1f us/launch-set  %6.1f TFLOP/s fp32-equivalent\n", name, taps, B, nstream,
           maxerr, maxerr / scale, ms / reps * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 768, C = 768, T = 936, B = 16, Tp = 8 * 128 + 2;
    const bool zero_data = getenv("ZERO") != nullptr;
    for (int taps : {1}) {
        std::vector<float> hw((size_t)taps * M * C), hx((size_t)B * C * T);
        srand(1);
        for (auto& v : hw) v = zero_data ? 0.f : ((rand() / (float)RAND_MAX) * 2 - 1) * 0.036f;
        for (auto& v : hx) v = zero_data ? 0.f : ((rand() / (float)RAND_MAX) * 2 - 1) * 1.7f;
        std::vector<unsigned short> wp((size_t)taps * (C / 8) * 3 * M * 8), xp((size_t)B * (C / 8) * 3 * Tp * 8, 0);
        unsigned short pl[3];
        for (int tap = 0; tap < taps; ++tap)
            for (int m = 0; m < M; ++m)
                for (int c = 0; c < C; ++c) {
                    split3(hw[((size_t)tap * M + m) * C + c], pl);
                    for (int p = 0; p < 3; ++p) wp[((((size_t)tap * (C / 8) + c / 8) * 3 + p) * M + m) * 8 + c % 8] = pl[p];
                }
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c)
                for (int t = 0; t < T; ++t) {
                    split3(hx[((size_t)b * C + c) * T + t], pl);
                    for (int p = 0; p < 3; ++p) xp[((((size_t)b * (C / 8) + c / 8) * 3 + p) * Tp + t + 1) * 8 + c % 8] = pl[p];
                }
        uint4 *Wp, *Xp; float* Y;
        (void)hipMalloc(&Wp, wp.size() * 2); (void)hipMalloc(&Xp, xp.size() * 2); (void)hipMalloc(&Y, (size_t)B * M * T * 4);
        (void)hipMemcpy(Wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(Xp, xp.data(), xp.size() * 2, hipMemcpyHostToDevice);
        run<2, 0, 0>(Wp, Xp, Y, M, C, taps, T, Tp, 16, 1, hw, hx, "current loop");
        runsr(Wp, Xp, Y, M, C, T, Tp, 16, 1, hw, hx, "strength-reduced loop");
        run<2, 0, 0>(Wp, Xp, Y, M, C, taps, T, Tp, 8, 1, hw, hx, "current loop");
        runsr(Wp, Xp, Y, M, C, T, Tp, 8, 1, hw, hx, "strength-reduced loop");
        run<2, 0, 0>(Wp, Xp, Y, M, C, taps, T, Tp, 8, 2, hw, hx, "current loop");
        runsr(Wp, Xp, Y, M, C, T, Tp, 8, 2, hw, hx, "strength-reduced loop");
        (void)hipFree(Wp); (void)hipFree(Xp); (void)hipFree(Y);
    }
    return 0;
}
