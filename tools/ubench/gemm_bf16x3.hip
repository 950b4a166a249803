// Prototype: fp32-accurate GEMM on bf16 MFMA by 3-way operand splitting (a = a0 + a1 + a2, bf16 each; 6 cross products).
//   Y[b][m][n] = sum_k W[m][k] * X[b][k][n]      W pre-split offline, X split on the fly while staging into LDS.
// Tile 128x128, 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16, BK = 16.
// LDS: operand rows are [row][7 chunks of 16 B] = planes (p0,p1,p2) x k-halves (0-7, 8-15) + 1 pad chunk -> conflict-free b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int BM = 128, BN = 128, BK = 16, PITCH = 112;   // bytes per operand row in LDS

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int NTERMS>   // 6 = bf16x3, 3 = bf16x2, 1 = plain bf16
__global__ __launch_bounds__(256) void gemm_split(const uint4* __restrict__ Wp, const float* __restrict__ X, float* __restrict__ Y, int K, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ws = smem;                            // [2][BM][PITCH]
    unsigned char* Xs = smem + 2 * BM * PITCH;           // [2][BN][PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
    const float* xb = X + (long long)b * K * T;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int nkb = K / BK;
    constexpr int WCH = BM * 7;                          // 16-B chunks of one W tile
    const int xn = tid & 127, xh = tid >> 7;             // X staging: column, k-half

    uint4 wreg[4];
    float xreg[8];
    auto load = [&](int kb) {
        const uint4* src = Wp + ((long long)kb * (gridDim.x * BM) + m0) * 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            if (c < WCH) wreg[i] = src[c];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) xreg[e] = xb[(long long)(kb * BK + 8 * xh + e) * T + n0 + xn];
    };
    auto store = [&](int buf) {
        uint4* wd = reinterpret_cast<uint4*>(Ws + buf * BM * PITCH);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            if (c < WCH) wd[c] = wreg[i];
        }
        unsigned short p0[8], p1[8], p2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = xreg[e];
            p0[e] = bf16_rne(v);
            const float r1 = v - bf16_to_f(p0[e]);
            p1[e] = bf16_rne(r1);
            const float r2 = r1 - bf16_to_f(p1[e]);
            p2[e] = bf16_rne(r2);
        }
        unsigned char* row = Xs + buf * BN * PITCH + xn * PITCH;
        auto pack = [](const unsigned short* p) {
            uint4 q;
            q.x = p[0] | ((unsigned)p[1] << 16); q.y = p[2] | ((unsigned)p[3] << 16);
            q.z = p[4] | ((unsigned)p[5] << 16); q.w = p[6] | ((unsigned)p[7] << 16);
            return q;
        };
        *reinterpret_cast<uint4*>(row + (0 * 2 + xh) * 16) = pack(p0);
        *reinterpret_cast<uint4*>(row + (1 * 2 + xh) * 16) = pack(p1);
        *reinterpret_cast<uint4*>(row + (2 * 2 + xh) * 16) = pack(p2);
    };

    f16v acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    load(0);
    store(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const bool has_next = kb + 1 < nkb;
        if (has_next) load(kb + 1);
        const unsigned char* wq = Ws + (kb & 1) * BM * PITCH + (wm0 + l31) * PITCH + lhi * 16;
        const unsigned char* xq = Xs + (kb & 1) * BN * PITCH + (wn0 + l31) * PITCH + lhi * 16;
        bf8 a[2][3], bb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[i][p] = *reinterpret_cast<const bf8*>(wq + i * 32 * PITCH + p * 32);
                bb[i][p] = *reinterpret_cast<const bf8*>(xq + i * 32 * PITCH + p * 32);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (NTERMS >= 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][2], acc[i][j], 0, 0, 0);
                }
                if (NTERMS >= 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][0], acc[i][j], 0, 0, 0);
            }
        if (has_next) store((kb + 1) & 1);
        __syncthreads();
    }
    float* yb = Y + (long long)b * gridDim.x * BM * T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                yb[(long long)row * T + n] = acc[i][j][r];
            }
}

static unsigned short h_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }

template <int NT>
void run(const uint4* Wp, const float* X, float* Y, int M, int K, int T, int B, const std::vector<float>& hw, const std::vector<float>& hx, const char* name) {
    dim3 grid(M / BM, T / BN, B);
    const size_t lds = 2 * BM * PITCH + 2 * BN * PITCH;
    hipLaunchKernelGGL(gemm_split<NT>, grid, dim3(256), lds, 0, Wp, X, Y, K, T);
    (void)hipDeviceSynchronize();
    std::vector<float> hy((size_t)M * T);
    (void)hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, scale = 0;
    for (int m = 0; m < M; m += 37)
        for (int n = 0; n < T; n += 53) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hw[(size_t)m * K + k] * (double)hx[(size_t)k * T + n];
            maxerr = fmax(maxerr, fabs(ref - hy[(size_t)m * T + n]));
            scale = fmax(scale, fabs(ref));
        }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split<NT>, grid, dim3(256), lds, 0, Wp, X, Y, K, T);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double fl = 2.0 * M * K * (double)T * B * reps;
    printf("%-22s max err %.3e (rel to max |y| %.2e)   %7.1f us/launch  %6.1f TFLOP/s (fp32-equivalent)\n", name, maxerr, maxerr / scale,
           ms / reps * 1e3, fl / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 768, K = 768, T = 1024, B = 16;
    std::vector<float> hw((size_t)M * K), hx((size_t)K * T * B);
    srand(1);
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.036f;
    for (auto& v : hx) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 1.7f;
    // pack W: [kb][m][7 chunks]: chunk = plane*2 + half; 8 bf16 per chunk
    const int nkb = K / BK;
    std::vector<unsigned short> wp((size_t)nkb * M * 7 * 8, 0);
    for (int kb = 0; kb < nkb; ++kb)
        for (int m = 0; m < M; ++m)
            for (int kk = 0; kk < 16; ++kk) {
                const float v = hw[(size_t)m * K + kb * 16 + kk];
                const unsigned short p0 = h_bf16(v); const float r1 = v - h_f(p0);
                const unsigned short p1 = h_bf16(r1); const float r2 = r1 - h_f(p1);
                const unsigned short p2 = h_bf16(r2);
                const unsigned short pl[3] = {p0, p1, p2};
                for (int p = 0; p < 3; ++p) wp[(((size_t)kb * M + m) * 7 + (p * 2 + kk / 8)) * 8 + (kk % 8)] = pl[p];
            }
    uint4* Wp; float *X, *Y;
    (void)hipMalloc(&Wp, wp.size() * 2); (void)hipMalloc(&X, hx.size() * 4); (void)hipMalloc(&Y, (size_t)B * M * T * 4);
    (void)hipMemcpy(Wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    run<6>(Wp, X, Y, M, K, T, B, hw, hx, "bf16x3 (6 products)");
    run<3>(Wp, X, Y, M, K, T, B, hw, hx, "bf16x2 (3 products)");
    run<1>(Wp, X, Y, M, K, T, B, hw, hx, "bf16   (1 product)");
    return 0;
}
