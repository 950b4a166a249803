// Tile-shape study of the split-precision (3 x bf16) conv GEMM at the bench's shapes (M = 768 / 2304, K = 768, B = 16, T = 936).
// Same operand layouts and K-loop structure as detail_tts_amd/csrc/conv_x3.hip (pre-split planes, LDS-DMA for both tiles, one
// barrier per 16-channel K-step, term-major MFMA order with the DMA pieces spread between MFMA groups), but the block tile
// (BM x BN), the wave grid (WGM x WGN) and hence the wave tile (MI x NJ MFMA 32x32x16 tiles) are template parameters.
// Throughput counts USEFUL flops (T = 936 columns), so the padding of the last N tile of every sample is charged to the variant.
//   hipcc --offload-arch=gfx950 -O3 -o bin/gemm_x3v gemm_x3v.hip && bin/gemm_x3v
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));

template <int NPL, int WGM, int WGN, int MI, int NJ, int MINW, int NSTG = 2, int KB = 1, int ABL = 0, int BUF = 0>
__global__ __launch_bounds__(WGM* WGN * 64, MINW) void gemm_x3v(const uint4* __restrict__ Wp, const uint4* __restrict__ Xp, float* __restrict__ Y,
                                                                int M, int C8, int Tp, int T, int ntn, float oscale) {
    constexpr int NW = WGM * WGN, BM = WGM * MI * 32, BN = WGN * NJ * 32;
    constexpr int NK = 2 * NPL;        // kinds per K-step: (plane, k-half)
    constexpr int ATILE = NK * BM * 16, BTILE = NK * BN * 16, SUB = ATILE + BTILE, STAGE = KB * SUB;       // a stage = KB 16-channel sub-steps
    constexpr int APIECES = NK * BM / 64, BPIECES = NK * BN / 64, SPIECES = APIECES + BPIECES, PIECES = KB * SPIECES;      // 1 KiB each
    constexpr int PPW = (PIECES + NW - 1) / NW;                                                 // pieces per wave (upper bound)
    constexpr int NT = NPL == 3 ? 6 : 3;
    constexpr int NMF = KB * NT * MI * NJ;                                                            // MFMAs per K-step per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    // 1-D grid, XCD-aware: M tiles of one (sample, N tile) adjacent
    const int mtiles = M / BM;
    int L;
    {
        const int nwg = gridDim.x, lin = blockIdx.x, xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    const int mt = L % mtiles, nb = L / mtiles, b = nb / ntn, n0 = (nb - b * ntn) * BN, m0 = mt * BM;
    const int c16n = C8 / 2;
    const uint4* wbase = Wp + m0 + lane;
    const uint4* xbase = Xp + (size_t)b * C8 * NPL * Tp + n0 + 1 + lane;
    // piece p (0 .. PIECES-1): p < APIECES -> W piece (kind = p / (BM/64), rh = p % (BM/64)); else X piece
    // BUF: buffer_load ... lds through two resource descriptors (W, X of this sample): per-lane offset lane*16 is loop-invariant, the
    // piece's offset is a scalar -> no vector address arithmetic per piece
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Wp + m0), (short)0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(Xp + (size_t)b * C8 * NPL * Tp + n0 + 1), (short)0, 0x7fffffff, 0x00020000);
    const int voff = lane * 16;
    auto issue = [&](int pp, int kstep, int stage) {            // kstep: index of the KB-block
        if (pp >= PIECES) return;
        const int sub = pp / SPIECES, p = pp - sub * SPIECES, c16 = kstep * KB + sub;
        unsigned char* sbase = smem + stage * STAGE + sub * SUB;
        if (p < APIECES) {
            const int kind = p / (BM / 64), rh = p % (BM / 64), pl = kind >> 1, h = kind & 1;
            if (BUF) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(sbase + kind * (BM * 16) + rh * 1024), 16, voff,
                                                         (((2 * c16 + h) * NPL + pl) * M + rh * 64) * 16, 0, 0);
            } else {
                const uint4* g = wbase + ((long long)(2 * c16 + h) * NPL + pl) * M + rh * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(sbase + kind * (BM * 16) + rh * 1024), 16, 0, 0);
            }
        } else {
            const int q = p - APIECES, kind = q / (BN / 64), rh = q % (BN / 64), pl = kind >> 1, h = kind & 1;
            if (BUF) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sbase + ATILE + kind * (BN * 16) + rh * 1024), 16, voff,
                                                         (((2 * c16 + h) * NPL + pl) * Tp + rh * 64) * 16, 0, 0);
            } else {
                const uint4* g = xbase + ((long long)(2 * c16 + h) * NPL + pl) * Tp + rh * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(sbase + ATILE + kind * (BN * 16) + rh * 1024), 16, 0, 0);
            }
        }
    };
    // BUF == 2 (round 4): VGPR staging instead of LDS-DMA - a piece is one global_load_dwordx4 into a register quad, stored to the LDS one
    // K-step later with ds_write_b128 (LDS double buffer + one register stage: the same two steps of prefetch distance as 3 LDS stages)
    auto g_addr = [&](int pp, int kstep) -> const uint4* {
        const int p = pp % SPIECES, c16 = kstep;
        if (p < APIECES) {
            const int kind = p / (BM / 64), rh = p % (BM / 64), pl = kind >> 1, h = kind & 1;
            return wbase + ((long long)(2 * c16 + h) * NPL + pl) * M + rh * 64;
        }
        const int q = p - APIECES, kind = q / (BN / 64), rh = q % (BN / 64), pl = kind >> 1, h = kind & 1;
        return xbase + ((long long)(2 * c16 + h) * NPL + pl) * Tp + rh * 64;
    };
    auto l_off = [&](int pp, int stage) -> int {
        const int p = pp % SPIECES;
        if (p < APIECES) {
            const int kind = p / (BM / 64), rh = p % (BM / 64);
            return stage * STAGE + kind * (BM * 16) + rh * 1024 + lane * 16;
        }
        const int q = p - APIECES, kind = q / (BN / 64), rh = q % (BN / 64);
        return stage * STAGE + ATILE + kind * (BN * 16) + rh * 1024 + lane * 16;
    };
    const int wm0 = (wave / WGN) * (MI * 32), wn0 = (wave % WGN) * (NJ * 32);
    f16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nkb = c16n / KB;
    if constexpr (BUF == 2) {
        static_assert(KB == 1 && NSTG == 2, "VGPR staging: LDS double buffer");
        uint4 sreg[PPW];
        auto pidx = [&](int i) { int p = wave + i * NW; return p >= PIECES ? wave : p; };
        // prologue: step 0 -> registers -> LDS stage 0; step 1 -> registers
#pragma unroll
        for (int i = 0; i < PPW; ++i) sreg[i] = *g_addr(pidx(i), 0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) *reinterpret_cast<uint4*>(smem + l_off(pidx(i), 0)) = sreg[i];
#pragma unroll
        for (int i = 0; i < PPW; ++i) sreg[i] = *g_addr(pidx(i), nkb > 1 ? 1 : 0);
        for (int ks = 0; ks < nkb; ++ks) {
            __syncthreads();                                   // the ds_writes of the previous step (stage ks % 2) are visible; stage (ks + 1) % 2 is free
            const int cur = ks & 1, nst = cur ^ 1, kx = ks + 2 < nkb ? ks + 2 : nkb - 1;
            const unsigned char* As = smem + cur * STAGE + lhi * (BM * 16);
            const unsigned char* Bs = smem + cur * STAGE + ATILE + lhi * (BN * 16);
            bf8 a[MI][NPL], bb[NJ][NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const bf8*>(As + p * (2 * BM * 16) + (wm0 + i * 32 + l31) * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bb[j][p] = *reinterpret_cast<const bf8*>(Bs + p * (2 * BN * 16) + (wn0 + j * 32 + l31) * 16);
            }
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
            int mf = 0, piece = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[i][TA[t]]), __builtin_bit_cast(hf8, bb[j][TB[t]]), acc[i][j], 0, 0, 0);
                        ++mf;
                        if (piece < PPW && mf * PPW >= (piece + 1) * NMF) {
                            __builtin_amdgcn_sched_barrier(0);
                            *reinterpret_cast<uint4*>(smem + l_off(pidx(piece), nst)) = sreg[piece];      // data of step ks + 1 -> LDS
                            sreg[piece] = *g_addr(pidx(piece), kx);                                        // data of step ks + 2 -> registers
                            __builtin_amdgcn_sched_barrier(0);
                            ++piece;
                        }
                    }
        }
    } else {
    // every wave issues exactly PPW load instructions per K-block (the ones past PIECES re-fetch piece 0 of the wave: harmless
    // duplicates), so a counted vmcnt is the same immediate for every wave
    auto issue_w = [&](int i, int kstep, int stage) { int p = wave + i * NW; if (p >= PIECES) p = wave; issue(p, kstep, stage); };
#pragma unroll
    for (int st = 0; st < NSTG - 1; ++st)
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_w(i, st < nkb ? st : nkb - 1, st);
    bf8 keep_a[MI][NPL], keep_b[NJ][NPL];
    for (int ks = 0; ks < nkb; ++ks) {
        // data of K-block ks was issued NSTG-1 blocks ago: at most (NSTG-2) newer blocks may stay in flight
        if (!(ABL & 2)) {
            if (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NSTG == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        const int cur = ks % NSTG, nst = (ks + NSTG - 1) % NSTG, kx = ks + NSTG - 1 < nkb ? ks + NSTG - 1 : nkb - 1;
        int mf = 0, piece = 0;
#pragma unroll
        for (int sub = 0; sub < KB; ++sub) {
            const unsigned char* As = smem + cur * STAGE + sub * SUB + lhi * (BM * 16);
            const unsigned char* Bs = smem + cur * STAGE + sub * SUB + ATILE + lhi * (BN * 16);
            bf8 a[MI][NPL], bb[NJ][NPL];
            if ((ABL & 4) && ks > 0) {          // ablation: no LDS fragment reads after the first step (operands stay in registers)
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[i][p] = keep_a[i][p];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bb[j][p] = keep_b[j][p];
                }
            } else {
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const bf8*>(As + p * (2 * BM * 16) + (wm0 + i * 32 + l31) * 16);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bb[j][p] = *reinterpret_cast<const bf8*>(Bs + p * (2 * BN * 16) + (wn0 + j * 32 + l31) * 16);
            }
            if (ABL & 4) {
#pragma unroll
                for (int p = 0; p < NPL; ++p) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) keep_a[i][p] = a[i][p];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) keep_b[j][p] = bb[j][p];
                }
            }
            }
            constexpr int TA[6] = {NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0, 0}, TB[6] = {0, 1, NPL == 3 ? 2 : 0, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (NPL == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[i][TA[t]]), __builtin_bit_cast(hf8, bb[j][TB[t]]), acc[i][j], 0, 0, 0);
                        ++mf;
                        if (piece < PPW && mf * PPW >= (piece + 1) * NMF) {       // this wave's PPW pieces spread evenly over the NMF MFMAs
                            __builtin_amdgcn_sched_barrier(0);
                            if (!(ABL & 1)) issue_w(piece, kx, nst);
                            __builtin_amdgcn_sched_barrier(0);
                            ++piece;
                        }
                    }
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* yb = Y + (long long)b * M * T;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                if (n < T) yb[(long long)row * T + n] = acc[i][j][r] * oscale;
            }
}

// ---- ping-pong variant: 256 x 192 tile, 8 waves = two groups of four (waves w and w + 4 share a SIMD).  The groups run half a K-step
// out of phase: while one group issues its 18 MFMAs of step s (M phase) the other reads its fragments of the next step from LDS and
// issues its share of the LDS-DMA two steps ahead (R phase); ONE barrier per phase.  3 LDS stages; every wave issues 4 DMA instructions
// per step and waits vmcnt(4) at the end of its R phase, so every piece issued at least one R phase earlier has landed at each barrier.
template <int NSTG>     // LDS stages: loads run NSTG - 1 steps ahead
__global__ __launch_bounds__(512, 2) void gemm_pp(const uint4* __restrict__ Wp, const uint4* __restrict__ Xp, float* __restrict__ Y, int M, int C8, int Tp,
                                                  int T, int ntn, float oscale) {
    constexpr int NPL = 2, NK = 4, BM = 256, BN = 192, MI = 2, NJ = 3, DIST = NSTG - 1;
    constexpr int ATILE = NK * BM * 16, BTILE = NK * BN * 16, STAGE = ATILE + BTILE;          // 16 + 12 KiB
    constexpr int APIECES = NK * BM / 64, BPIECES = NK * BN / 64, PIECES = APIECES + BPIECES;  // 16 + 12 = 28
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int grp = wave >> 2, wq = wave & 3;        // group (phase parity), index inside the group
    const int mtiles = M / BM;
    int L;
    {
        const int nwg = gridDim.x, lin = blockIdx.x, xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    const int mt = L % mtiles, nb = L / mtiles, b = nb / ntn, n0 = (nb - b * ntn) * BN, m0 = mt * BM;
    const int nks = C8 / 2;
    const uint4* wbase = Wp + m0 + lane;
    const uint4* xbase = Xp + (size_t)b * C8 * NPL * Tp + n0 + 1 + lane;
    auto issue = [&](int i, int ks) {               // this wave's i-th piece (0..3) of step ks (clamped)
        const int c16 = ks < nks ? ks : nks - 1, stage = ks % NSTG;
        int p = wave + 8 * i;
        if (p >= PIECES) p = wave;                  // duplicates keep the instruction count uniform
        if (p < APIECES) {
            const int kind = p >> 2, rh = p & 3, pl = kind >> 1, h = kind & 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + ((long long)(2 * c16 + h) * NPL + pl) * M + rh * 64),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + kind * (BM * 16) + rh * 1024), 16, 0, 0);
        } else {
            const int q = p - APIECES, kind = q / 3, rh = q - kind * 3, pl = kind >> 1, h = kind & 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + ((long long)(2 * c16 + h) * NPL + pl) * Tp + rh * 64),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + ATILE + kind * (BN * 16) + rh * 1024), 16, 0, 0);
        }
    };
    // wave grid 4 (M) x 2 (N): the two waves of a SIMD (w, w + 4) take different row blocks
    const int wr = (wq & 1) * 2 + grp, wc = wq >> 1;
    const int wm0 = wr * 64, wn0 = wc * 96;
    f16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int st = 0; st < DIST; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i) issue(i, st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf8 a[MI][NPL], bb[NJ][NPL];
    auto rphase = [&](int ks) {                     // fragments of step ks, then this wave's DMA share two steps ahead
        const unsigned char* As = smem + (ks % NSTG) * STAGE + lhi * (BM * 16);
        const unsigned char* Bs = smem + (ks % NSTG) * STAGE + ATILE + lhi * (BN * 16);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i][p] = *reinterpret_cast<const bf8*>(As + p * (2 * BM * 16) + (wm0 + i * 32 + l31) * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bb[j][p] = *reinterpret_cast<const bf8*>(Bs + p * (2 * BN * 16) + (wn0 + j * 32 + l31) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) issue(i, ks + DIST);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DIST - 1)) : "memory");       // all but the newest DIST - 1 steps' pieces have landed
    };
    auto mphase = [&]() {
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[i][TA[t]]), __builtin_bit_cast(hf8, bb[j][TB[t]]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // group 0: R(0) | M R(1) | ... ; group 1 runs the same sequence one phase later.  Two straight-line loops (a shared loop with a
    // per-phase branch makes the compiler copy all 96 accumulator registers around every M phase).
    auto step = [&](int ks) {
        rphase(ks);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        mphase();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    if (grp == 0) {
        for (int ks = 0; ks < nks; ++ks) step(ks);
        __builtin_amdgcn_s_barrier();
    } else {
        __builtin_amdgcn_s_barrier();
        for (int ks = 0; ks < nks; ++ks) step(ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* yb = Y + (long long)b * M * T;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, n = n0 + wn0 + j * 32 + l31;
                if (n < T) yb[(long long)row * T + n] = acc[i][j][r] * oscale;
            }
}

static unsigned short h_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float h_f(unsigned short h) { unsigned u = ((unsigned)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short h_f16(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h_h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static void split2h(float v, unsigned short* p) { p[0] = h_f16(v); p[1] = h_f16(v - h_h2f(p[0])); }
static void split3(float v, unsigned short* p) {
    p[0] = h_bf16(v); const float r1 = v - h_f(p[0]);
    p[1] = h_bf16(r1); const float r2 = r1 - h_f(p[1]);
    p[2] = h_bf16(r2);
}

template <int NPL, int WGM, int WGN, int MI, int NJ, int MINW, int NSTG = 2, int KB = 1, int ABL = 0, int BUF = 0>
void run(const uint4* Wp, const uint4* Xp, float* Y, int M, int C, int T, int Tp, int B, const std::vector<float>& hw, const std::vector<float>& hx,
         const char* name, float oscale) {
    constexpr int BM = WGM * MI * 32, BN = WGN * NJ * 32;
    if (M % BM) { printf("%-34s M %d not a multiple of %d\n", name, M, BM); return; }
    const int ntn = (T + BN - 1) / BN;
    if (ntn * BN + 2 > Tp) { printf("%-34s Tp too small\n", name); return; }
    const dim3 grid((M / BM) * ntn * B);
    const size_t lds = (size_t)NSTG * KB * 2 * NPL * 16 * (BM + BN);
    auto k = gemm_x3v<NPL, WGM, WGN, MI, NJ, MINW, NSTG, KB, ABL, BUF>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, WGM * WGN * 64, lds);
    hipLaunchKernelGGL(k, grid, dim3(WGM * WGN * 64), lds, 0, Wp, Xp, Y, M, C / 8, Tp, T, ntn, oscale);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-34s launch failed\n", name); return; }
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, tot = 0.f;
    const int reps = 5, rounds = 4;
    for (int r = 0; r < rounds; ++r) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(WGM * WGN * 64), lds, 0, Wp, Xp, Y, M, C / 8, Tp, T, ntn, oscale);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms / reps);
        tot += ms / reps;
    }
    const double fl = 2.0 * M * C * (double)T * B;
    printf("%-26s st%d kb%d M %4d  tile %3dx%3d  %4d WGs (%d/CU, lds %3zu KiB)  rel err %.1e  %7.1f us (best %7.1f)  %6.1f TF-eq (best %6.1f)  frac %.3f\n", name, NSTG, KB,
           M, BM, BN, grid.x, occ, lds >> 10, maxerr / scale, tot / rounds * 1e3, best * 1e3, fl / (tot / rounds * 1e-3) / 1e12,
           fl / (best * 1e-3) / 1e12, (NPL == 3 ? 6 : 3) * fl / (best * 1e-3) / 1e12 / 2500.0);
}

template <int NSTG>
void run_pp(const uint4* Wp, const uint4* Xp, float* Y, int M, int C, int T, int Tp, int B, const std::vector<float>& hw, const std::vector<float>& hx,
            const char* name, float oscale) {
    constexpr int BM = 256, BN = 192;
    if (M % BM) { printf("%-26s M %d not a multiple of %d\n", name, M, BM); return; }
    const int ntn = (T + BN - 1) / BN;
    const dim3 grid((M / BM) * ntn * B);
    const size_t lds = (size_t)NSTG * 4 * 16 * (BM + BN);
    auto k = gemm_pp<NSTG>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, Wp, Xp, Y, M, C / 8, Tp, T, ntn, oscale);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-26s launch failed\n", name); return; }
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
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, tot = 0.f;
    const int reps = 5, rounds = 4;
    for (int r = 0; r < rounds; ++r) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(512), lds, 0, Wp, Xp, Y, M, C / 8, Tp, T, ntn, oscale);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms / reps);
        tot += ms / reps;
    }
    const double fl = 2.0 * M * C * (double)T * B;
    printf("%-26s ping-pong st%d M %4d  tile 256x192  %4d WGs  rel err %.1e  %7.1f us (best %7.1f)  %6.1f TF-eq (best %6.1f)  frac %.3f\n", name, NSTG, M, grid.x,
           maxerr / scale, tot / rounds * 1e3, best * 1e3, fl / (tot / rounds * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12, 3 * fl / (best * 1e-3) / 1e12 / 2500.0);
}

template <int NPL>
void suite(int M, int C, int T, int B, int Tp, const std::vector<float>& hw, const std::vector<float>& hx) {
    const float sw = NPL == 2 ? 64.f : 1.f, sx = NPL == 2 ? 16.f : 1.f;
    std::vector<unsigned short> wp((size_t)(C / 8) * NPL * M * 8), xp((size_t)B * (C / 8) * NPL * Tp * 8, 0);
    unsigned short pl[3];
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < C; ++c) {
            if (NPL == 3) split3(hw[(size_t)m * C + c], pl); else split2h(hw[(size_t)m * C + c] * sw, pl);
            for (int p = 0; p < NPL; ++p) wp[((((size_t)c / 8) * NPL + p) * M + m) * 8 + c % 8] = pl[p];
        }
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int t = 0; t < T; ++t) {
                if (NPL == 3) split3(hx[((size_t)b * C + c) * T + t], pl); else split2h(hx[((size_t)b * C + c) * T + t] * sx, pl);
                for (int p = 0; p < NPL; ++p) xp[((((size_t)b * (C / 8) + c / 8) * NPL + p) * Tp + t + 1) * 8 + c % 8] = pl[p];
            }
    uint4 *Wp, *Xp; float* Y;
    (void)hipMalloc(&Wp, wp.size() * 2); (void)hipMalloc(&Xp, xp.size() * 2); (void)hipMalloc(&Y, (size_t)B * M * T * 4);
    (void)hipMemcpy(Wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(Xp, xp.data(), xp.size() * 2, hipMemcpyHostToDevice);
    const float os = 1.f / (sw * sx);
    printf("---- %s, M = %d\n", NPL == 3 ? "3 x bf16 planes, 6 products" : "2 x fp16 planes, 3 products", M);
    if (getenv("UB_SMALL")) {      // under-filled launches (B = 8, M = 768: 240 workgroups of 128 x 192): do smaller tiles pay?
        for (int rep = 0; rep < 2; ++rep) {
            run<NPL, 2, 2, 2, 3, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 4 waves (64x96)", os);
            run<NPL, 2, 2, 2, 2, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x128 4 waves (64x64)", os);
            run<NPL, 2, 2, 1, 3, 4>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "64x192 4 waves (32x96)", os);
            run<NPL, 2, 1, 2, 3, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x96 2 waves (64x96)", os);
            run<NPL, 4, 1, 1, 3, 4>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x96 4 waves (32x96)", os);
            run<NPL, 2, 1, 1, 3, 4>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "64x96 2 waves (32x96)", os);
            run<NPL, 1, 2, 2, 3, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "64x192 2 waves (64x96)", os);
        }
        (void)hipFree(Wp); (void)hipFree(Xp); (void)hipFree(Y);
        return;
    }
    if (getenv("UB_BIG")) {        // round 4: 4 waves with 128 x 96 wave tiles (one wave per SIMD) against the product's tile, 3 LDS stages
        for (int rep = 0; rep < 2; ++rep) {
            run<NPL, 2, 2, 2, 3, 2, 3>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 4w (64x96) st3", os);
            run<NPL, 2, 2, 2, 3, 2, 2, 1, 0, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 4w VGPR-staged", os);
            run<NPL, 4, 2, 2, 3, 1, 2, 1, 0, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 8w VGPR-staged", os);
            run<NPL, 2, 2, 4, 3, 1, 3>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 4w (128x96) st3", os);
            run<NPL, 2, 2, 4, 3, 1, 4>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 4w (128x96) st4", os);
            run<NPL, 2, 2, 3, 3, 1, 3>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "192x192 4w (96x96) st3", os);
            run<NPL, 4, 2, 2, 3, 1, 3>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 8w (64x96) st3", os);
        }
        (void)hipFree(Wp); (void)hipFree(Xp); (void)hipFree(Y);
        return;
    }
    for (int rep = 0; rep < 2; ++rep) {
        if (NPL == 2) { run_pp<3>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 8 waves", os); run_pp<4>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 8 waves", os); run_pp<5>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 8 waves", os); }
        run<NPL, 2, 2, 2, 3, 2>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 global_load_lds", os);
        run<NPL, 2, 2, 2, 3, 2, 2, 1, 0, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 buffer_load lds", os);
        run<NPL, 2, 2, 2, 3, 2, 3, 1, 0, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 buffer st3", os);
        run<NPL, 2, 2, 2, 3, 2, 2, 1, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "128x192 no DMA", os);
        run<NPL, 4, 2, 2, 3, 1, 3, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 st3 global", os);
        run<NPL, 4, 2, 2, 3, 1, 3, 1, 0, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 st3 buffer", os);
        run<NPL, 4, 2, 2, 3, 1, 2, 1, 0, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "256x192 st2 buffer", os);
        run<NPL, 4, 2, 3, 3, 1, 3, 1, 0, 1>(Wp, Xp, Y, M, C, T, Tp, B, hw, hx, "384x192 st3 buffer", os);
    }
    (void)hipFree(Wp); (void)hipFree(Xp); (void)hipFree(Y);
}

int main() {
    const int C = getenv("UB_C") ? atoi(getenv("UB_C")) : 768, T = 936, B = getenv("UB_B") ? atoi(getenv("UB_B")) : 16, Tp = 1024 + 2 + 256;
    for (int M : {768, 2304}) {
        std::vector<float> hw((size_t)M * C), hx((size_t)B * C * T);
        srand(1);
        for (auto& v : hw) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.036f;
        for (auto& v : hx) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 1.7f;
        suite<2>(M, C, T, B, Tp, hw, hx);
        if (!getenv("SKIP3")) suite<3>(M, C, T, B, Tp, hw, hx);
    }
    return 0;
}
