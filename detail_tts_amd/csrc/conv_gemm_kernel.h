// fp32 MFMA conv-as-GEMM kernel template for gfx950 (see conv_gemm.h / conv_gemm.hip for the design notes).
#pragma once
#include "conv_gemm.h"
#include "prof.h"

namespace dtts {

__device__ __forceinline__ float silu_fast(float v) { return v * __frcp_rn(1.f + __expf(-v)); }

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int XCOLS = 3;       // column iterations of 64 lanes -> XW <= 192
constexpr int XW_MAX = 64 * XCOLS;

// PRO: 0 raw input | 1 affine + SiLU | 2 affine only | 3 leaky-relu (no affine) | 4 SiLU (no affine)
// EPI: 0 linear (bias, per-sample rows, scale, residual, polyphase scatter) | 1 + generic activation | 2 gated pair
// KWT: 0 = generic (runtime KW / stride / dilation) | 1, 3 = compile-time tap count with stride 1, dilation 1 (the diffusion convs)
template <int BM, int BN, int WGM, int WGN, int BK, int PRO, int EPI, int KWT>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    const int KW = KWT > 0 ? KWT : p.KW;
    const int stride = KWT > 0 ? 1 : p.stride;
    const int dil = KWT > 0 ? 1 : p.dil;
    constexpr int XR = BK / 4;                        // channel rows staged per wave
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int WV4 = (BK * BM / 4) / 256;          // float4 weight loads per thread
    static_assert((BK * BM / 4) % 256 == 0 || (BK * BM / 4) < 256, "w tile");
    constexpr int WLOADS = WV4 > 0 ? WV4 : 1;

    extern __shared__ float smem[];
    const int XW = (BN - 1) * stride + (KW - 1) * dil + 1;
    const int XWP = XW + 1;                            // row pitch
    float* Ws = smem;                                  // [2][BK][BM]
    float* Xs = smem + 2 * BK * BM;                    // [2][BK][XWP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // 1-D grid, XCD-aware: all M-tiles of one (sample, N-tile) are adjacent logical ids, so the blocks that share an input
    // tile run on the same XCD and hit its L2 (the weights, 2-7 MB, are read by every XCD anyway)
    const int mtiles = p.CoutP / BM, ntiles = (p.Nout + BN - 1) / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % mtiles, nb = L / mtiles;
    const int b = nb / ntiles;
    const int m0 = mt * BM;
    const int n0 = (nb - b * ntiles) * BN;
    const int nvalid = p.len_out ? p.len_out[b] : p.Nout;
    if (n0 >= nvalid) return;
    const int lin = p.len_in ? p.len_in[b] : p.Tin;

    const float* xb = p.x + (long long)(p.x_bidx ? p.x_bidx[b] : b) * p.x_bs;
    const float* ab = p.pro_ab ? p.pro_ab + (long long)b * p.Cin * 2 : nullptr;
    const int tin0 = n0 * stride - p.pad;

    const int nCb = p.CinP / BK;
    const int S = nCb * KW;

    float xreg[XR * XCOLS];

    // ---- branch-free staging: every load is unconditional on a clamped address, masking happens by select at store
    // time, so all loads of a K-step are in flight together and land behind the MFMA section.
    const int lin_c = lin > 0 ? lin - 1 : 0;
    int xoff[XCOLS];          // clamped time index of this lane's columns
    unsigned xok = 0;         // bit (c): column valid (inside the tile and inside [0, len))
#pragma unroll
    for (int c = 0; c < XCOLS; ++c) {
        const int col = lane + 64 * c;
        const int t = tin0 + col;
        const bool ok = (col < XW) && (t >= 0) && (t < lin);
        xok |= (ok ? 1u : 0u) << c;
        xoff[c] = min(max(t, 0), lin_c);
    }
    // weight tile: LDS-DMA (global_load_lds_dwordx4): no VGPR staging, no ds_write.  The tile is [BK][BM] row-major in
    // LDS == lane-linear: float4 #idx of the tile goes to byte offset 16*idx, one wave-instruction fills 1 KiB.
    auto load_w = [&](int cb, int tap, int buf) {
        const float* wp = p.w + ((long long)(tap * p.CinP + cb * BK)) * p.CoutP + m0;
        float* lbase = Ws + buf * BK * BM;
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int idx = tid + i * 256;
            if (WV4 > 0 || (wave * 64 < BK * BM / 4)) {
                const int row = idx / (BM / 4), c4 = idx - row * (BM / 4);
                const float* g = wp + (long long)row * p.CoutP + c4 * 4;
                float* l = lbase + (wave * 64 + i * 256) * 4;      // wave-uniform base; hardware adds lane*16
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)l, 16, 0, 0);
            }
        }
    };
    // each wave stages 4 of the 16 channel rows; lanes stride over columns
    float pa[XR], pd[XR];
    unsigned rok = 0;
    auto load_x = [&](int cb) {
        rok = 0;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int ci = cb * BK + wave * XR + r;
            const bool cok = ci < p.Cin;
            const int cic = cok ? ci : p.Cin - 1;
            rok |= (cok ? 1u : 0u) << r;
            if (PRO == 1 || PRO == 2) { pa[r] = ab[cic * 2]; pd[r] = ab[cic * 2 + 1]; }
            const float* xr = xb + (long long)cic * p.x_cs;
#pragma unroll
            for (int c = 0; c < XCOLS; ++c)
                if (c == 0 || 64 * c < XW) xreg[r * XCOLS + c] = xr[xoff[c]];      // wave-uniform guard
        }
    };
    auto store_x = [&](int buf) {
        float* dst = Xs + buf * BK * XWP + (wave * XR) * XWP + lane;
#pragma unroll
        for (int r = 0; r < XR; ++r)
#pragma unroll
            for (int c = 0; c < XCOLS; ++c) {
                if (c > 0 && 64 * c >= XWP) continue;                          // wave-uniform
                float v = xreg[r * XCOLS + c];
                if (PRO == 1) { v = pa[r] * v + pd[r]; v = silu_fast(v); }
                else if (PRO == 2) v = pa[r] * v + pd[r];
                else if (PRO == 3) v = v >= 0.f ? v : v * p.pro_slope;
                else if (PRO == 4) v = silu_fast(v);
                const bool ok = ((xok >> c) & 1u) && ((rok >> r) & 1u);
                if (lane + 64 * c < XWP) dst[r * XWP + 64 * c] = ok ? v : 0.f;
            }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    load_w(0, 0, 0);
    load_x(0);
    store_x(0);
    __syncthreads();

    int cb = 0, tap = 0;
    for (int s = 0; s < S; ++s) {
        const bool has_next = (s + 1) < S;
        int ntap = tap + 1, ncb = cb;
        if (ntap == KW) { ntap = 0; ncb = cb + 1; }
        const bool newx = has_next && (ncb != cb);
        if (has_next && !(p.ablate & 1)) load_w(ncb, ntap, (s + 1) & 1);
        if (newx && !(p.ablate & 1)) load_x(ncb);

        const float* wq = Ws + (s & 1) * BK * BM + lhi * BM + wm0 + l31;
        const float* xq = Xs + (cb & 1) * BK * XWP + lhi * XWP + (wn0 + l31) * stride + tap * dil;
        const int xw2 = 2 * XWP, nstep = 32 * stride;
        float af[BK / 2][TM], bf[BK / 2][TN];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[kk][i] = wq[kk * 2 * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[kk][j] = xq[kk * xw2 + j * nstep];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][i], bf[kk][j], acc[i][j], 0, 0, 0);

        if (newx && !(p.ablate & 2)) store_x(ncb & 1);
        if (!(p.ablate & 4)) __syncthreads();
        cb = ncb;
        tap = ntap;
    }

    // ---- epilogue.  C layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* yb = p.y + (long long)b * p.y_bs;
    const float* rb = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
    const float* bad = p.badd ? p.badd + (long long)b * p.badd_bs : nullptr;
    const int cpp = p.Cout / p.phases;   // real rows per phase
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            const bool nok = n < nvalid;
            if (EPI != 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (!nok || row >= p.Cout) continue;
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[row];
                    if (bad) v += bad[row];
                    if (EPI == 1) v = act_apply(v, p.epi_act, p.epi_slope);
                    v *= p.out_scale;
                    int co = row, t = n;
                    if (p.phases > 1) { const int ph = row / cpp; co = row - ph * cpp; t = n * p.phases + ph; }
                    if (rb) v += p.res_scale * rb[(long long)co * p.res_cs + t];
                    yb[(long long)co * p.y_cs + t] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;   // even
                    if (!nok || row >= p.Cout) continue;
                    float va = acc[i][j][r], vb = acc[i][j][r + 1];
                    if (p.bias) { va += p.bias[row]; vb += p.bias[row + 1]; }
                    if (bad) { va += bad[row]; vb += bad[row + 1]; }
                    float v = (p.gate == GATE_TANH_SIGMOID ? tanhf(va) : va) * sigmoidf_(vb);
                    v = act_apply(v, p.epi_act, p.epi_slope) * p.out_scale;
                    const int co = row >> 1;
                    if (rb) v += p.res_scale * rb[(long long)co * p.res_cs + n];
                    yb[(long long)co * p.y_cs + n] = v;
                }
            }
        }
    }
}


template <int BM, int BN, int WGM, int WGN, int BK>
void launch_conv_tile(const ConvParams& p, hipStream_t stream, const char* tag);

// one translation unit per tile shape instantiates the PRO x EPI grid
#define DTTS_INSTANTIATE_CONV_TILE(BM, BN, WGM, WGN, BK, FAST)                                                                      \
    template <int PRO, int EPI>                                                                                           \
    static void launch_pe_##BM##_##BN##_##BK(const ConvParams& p, dim3 grid, size_t lds, hipStream_t stream) {                   \
        static size_t lds_attr = 64 * 1024;                                                                               \
        if (lds > lds_attr) {                                                                                             \
            DTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<BM, BN, WGM, WGN, BK, PRO, EPI, 0>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
            lds_attr = 160 * 1024;                                                                                        \
        }                                                                                                                 \
        constexpr bool FASTPATH = (FAST) && EPI <= (BM == 64 && BN == 64 ? 1 : 0) && PRO <= 2;   /* the small-launch tile also takes its activation epilogue (GPT c_fc) on the compile-time-tap path */                                                         \
        if (FASTPATH && p.stride == 1 && p.dil == 1 && p.KW == 1 && lds <= 64 * 1024)                                    \
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WGM, WGN, BK, PRO, EPI, FASTPATH ? 1 : 0>), grid, dim3(256), lds, stream, p); \
        else if (FASTPATH && p.stride == 1 && p.dil == 1 && p.KW == 3 && lds <= 64 * 1024)                               \
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WGM, WGN, BK, PRO, EPI, FASTPATH ? 3 : 0>), grid, dim3(256), lds, stream, p); \
        else                                                                                                              \
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WGM, WGN, BK, PRO, EPI, 0>), grid, dim3(256), lds, stream, p);   \
    }                                                                                                                     \
    template <int PRO>                                                                                                    \
    static void launch_p_##BM##_##BN##_##BK(const ConvParams& p, int epi, dim3 grid, size_t lds, hipStream_t stream) {           \
        if (epi == 0) launch_pe_##BM##_##BN##_##BK<PRO, 0>(p, grid, lds, stream);                                                \
        else if (epi == 1) launch_pe_##BM##_##BN##_##BK<PRO, 1>(p, grid, lds, stream);                                           \
        else launch_pe_##BM##_##BN##_##BK<PRO, 2>(p, grid, lds, stream);                                                         \
    }                                                                                                                     \
    template <>                                                                                                           \
    void launch_conv_tile<BM, BN, WGM, WGN, BK>(const ConvParams& p, hipStream_t stream, const char* tag) {                   \
        const int XW = (BN - 1) * p.stride + (p.KW - 1) * p.dil + 1;                                                      \
        DTTS_REQUIRE(XW <= XW_MAX, "conv input tile too wide for the staging registers");                                 \
        DTTS_REQUIRE(p.CoutP % BM == 0 && p.CinP % BK == 0, "packed weight padding");                                     \
        const size_t lds = sizeof(float) * (2 * BK * BM + 2 * BK * (XW + 1));                                             \
        DTTS_REQUIRE(lds <= 160 * 1024, "conv LDS tile");                                                                 \
        dim3 grid((p.CoutP / BM) * cdiv(p.Nout, BN) * p.B);                                                               \
        const int pro = p.pro_ab ? (p.pro_act == ACT_SILU ? 1 : 2)                                                        \
                                 : (p.pro_act == ACT_LRELU ? 3 : (p.pro_act == ACT_SILU ? 4 : 0));                        \
        DTTS_REQUIRE(p.pro_ab ? (p.pro_act == ACT_SILU || p.pro_act == ACT_NONE)                                          \
                              : (p.pro_act == ACT_LRELU || p.pro_act == ACT_NONE || p.pro_act == ACT_SILU),               \
                     "unsupported conv prologue");                                                                        \
        const int epi = p.gate != GATE_NONE ? 2 : (p.epi_act != ACT_NONE ? 1 : 0);                                        \
        const double cols = (double)p.B * p.Nout;                                                                         \
        const double flops = 2.0 * p.Cout * p.Cin * p.KW * cols;                                                          \
        const double bytes = 4.0 * (cols * p.stride * p.Cin + cols * p.Cout * (p.res ? 2.0 : 1.0) + (double)p.Cout * p.Cin * p.KW); \
        {                                                                                                                 \
            ProfScope ps(tag, flops, bytes, stream);                                                                      \
            if (pro == 0) launch_p_##BM##_##BN##_##BK<0>(p, epi, grid, lds, stream);                                             \
            else if (pro == 1) launch_p_##BM##_##BN##_##BK<1>(p, epi, grid, lds, stream);                                        \
            else if (pro == 2) launch_p_##BM##_##BN##_##BK<2>(p, epi, grid, lds, stream);                                        \
            else if (pro == 3) launch_p_##BM##_##BN##_##BK<3>(p, epi, grid, lds, stream);                                        \
            else launch_p_##BM##_##BN##_##BK<4>(p, epi, grid, lds, stream);                                                      \
        }                                                                                                                 \
        DTTS_CHECK_HIP(hipGetLastError());                                                                                \
    }

}  // namespace dtts
