// fp32 MFMA conv-as-GEMM for gfx950 (see conv_gemm.h).
//
// Tiling: a 256-thread workgroup (4 waves) owns a BM x BN tile of y[b]; each wave owns a
// (BM/WGM) x (BN/WGN) sub-tile built from 32x32 v_mfma_f32_32x32x2_f32 tiles (exact fp32,
// 64 FLOP/clk/SIMD — the only fp32 matrix path on gfx950).  K runs over (ci-block of 16, tap):
// the input tile [16][XW] (XW = (BN-1)*stride + (KW-1)*dil + 1, halo included) is staged in
// LDS ONCE per ci-block and re-used by every tap through a shifted read; the weight tile
// [16][BM] is staged per (ci-block, tap).  Both tiles are K-major in LDS so every MFMA operand
// read is a conflict-free ds_read_b32 of 32 consecutive floats per lane-half.  Global->LDS goes
// through registers (double-buffered LDS, one barrier per K-step) because the input path applies
// the fused prologue (GroupNorm affine + SiLU / leaky-relu, zero padding, per-sample length).
#include "conv_gemm.h"
#include "prof.h"

namespace dtts {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int XCOLS = 3;       // column iterations of 64 lanes -> XW <= 192
constexpr int XW_MAX = 64 * XCOLS;

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int WV4 = (BK * BM / 4) / 256;          // float4 weight loads per thread
    static_assert((BK * BM / 4) % 256 == 0 || (BK * BM / 4) < 256, "w tile");
    constexpr int WLOADS = WV4 > 0 ? WV4 : 1;

    extern __shared__ float smem[];
    const int XW = (BN - 1) * p.stride + (p.KW - 1) * p.dil + 1;
    const int XWP = XW + 1;                            // row pitch
    float* Ws = smem;                                  // [2][BK][BM]
    float* Xs = smem + 2 * BK * BM;                    // [2][BK][XWP]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int nvalid = p.len_out ? p.len_out[b] : p.Nout;
    if (n0 >= nvalid) return;
    const int lin = p.len_in ? p.len_in[b] : p.Tin;

    const float* xb = p.x + (long long)b * p.x_bs;
    const float* ab = p.pro_ab ? p.pro_ab + (long long)b * p.Cin * 2 : nullptr;
    const int tin0 = n0 * p.stride - p.pad;

    const int nCb = p.CinP / BK;
    const int S = nCb * p.KW;

    float4 wreg[WLOADS];
    float xreg[4 * XCOLS];

    auto load_w = [&](int s) {
        const int cb = s / p.KW, tap = s - cb * p.KW;
        const float* wp = p.w + ((long long)(tap * p.CinP + cb * BK)) * p.CoutP + m0;
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            int idx = tid + i * 256;
            if (idx < BK * BM / 4) {
                int row = idx / (BM / 4), c4 = idx - row * (BM / 4);
                wreg[i] = *reinterpret_cast<const float4*>(wp + (long long)row * p.CoutP + c4 * 4);
            }
        }
    };
    auto store_w = [&](int buf) {
        float* dst = Ws + buf * BK * BM;
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            int idx = tid + i * 256;
            if (idx < BK * BM / 4) *reinterpret_cast<float4*>(dst + idx * 4) = wreg[i];
        }
    };
    // each wave stages 4 of the 16 channel rows; lanes stride over columns
    auto load_x = [&](int cb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = cb * BK + wave * 4 + r;
            float a = 1.f, d = 0.f;
            const bool cok = ci < p.Cin;
            if (ab && cok) { a = ab[ci * 2]; d = ab[ci * 2 + 1]; }
            const float* xr = xb + (long long)ci * p.x_cs;
#pragma unroll
            for (int c = 0; c < XCOLS; ++c) {
                const int col = lane + 64 * c;
                const int t = tin0 + col;
                float v = 0.f;
                if (cok && col < XW && t >= 0 && t < lin) {
                    v = xr[t];
                    if (ab) v = a * v + d;
                    v = act_apply(v, p.pro_act, p.pro_slope);
                }
                xreg[r * XCOLS + c] = v;
            }
        }
    };
    auto store_x = [&](int buf) {
        float* dst = Xs + buf * BK * XWP;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < XCOLS; ++c) {
                const int col = lane + 64 * c;
                if (col < XW) dst[(wave * 4 + r) * XWP + col] = xreg[r * XCOLS + c];
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

    load_w(0);
    load_x(0);
    store_w(0);
    store_x(0);
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const int cb = s / p.KW, tap = s - cb * p.KW;
        const bool has_next = (s + 1) < S;
        const int ncb = (s + 1) / p.KW;
        const bool newx = has_next && (ncb != cb);
        if (has_next) load_w(s + 1);
        if (newx) load_x(ncb);

        const float* wsb = Ws + (s & 1) * BK * BM + wm0 + l31;
        const float* xsb = Xs + (cb & 1) * BK * XWP + (wn0 + l31) * p.stride + tap * p.dil;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float a[TM], bv[TN];
            const int krow = kk * 2 + lhi;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = wsb[krow * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = xsb[krow * XWP + j * 32 * p.stride];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
        }

        if (has_next) store_w((s + 1) & 1);
        if (newx) store_x(ncb & 1);
        __syncthreads();
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
            if (p.gate == GATE_NONE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (!nok || row >= p.Cout) continue;
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[row];
                    if (bad) v += bad[row];
                    v = act_apply(v, p.epi_act, p.epi_slope) * p.out_scale;
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

template <int BM, int BN, int WGM, int WGN>
static void launch_cfg(const ConvParams& p, hipStream_t stream, const char* tag) {
    const int XW = (BN - 1) * p.stride + (p.KW - 1) * p.dil + 1;
    DTTS_REQUIRE(XW <= XW_MAX, "conv input tile too wide for the staging registers");
    DTTS_REQUIRE(p.CoutP % BM == 0 && p.CinP % BK == 0, "packed weight padding");
    const size_t lds = sizeof(float) * (2 * BK * BM + 2 * BK * (XW + 1));
    dim3 grid(p.CoutP / BM, cdiv(p.Nout, BN), p.B);
    DTTS_REQUIRE(lds <= 64 * 1024, "conv LDS tile");
    {
        // algorithmic work of this launch: 2*rows*Cin*KW MACs per output column; bytes = x + y (+res) + weights once
        const double cols = (double)p.B * p.Nout;
        const double flops = 2.0 * p.Cout * p.Cin * p.KW * cols;
        const double bytes = 4.0 * (cols * p.stride * p.Cin + cols * p.Cout * (p.res ? 2.0 : 1.0) + (double)p.Cout * p.Cin * p.KW);
        ProfScope ps(tag, flops, bytes, stream);
        hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WGM, WGN>), grid, dim3(256), lds, stream, p);
    }
    DTTS_CHECK_HIP(hipGetLastError());
}

void launch_conv_gemm(const ConvParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.B > 0 && p.Nout > 0 && p.Cout > 0 && p.Cin > 0, "empty conv");
    DTTS_REQUIRE(p.x && p.w && p.y, "null pointer");
    DTTS_REQUIRE(p.gate == GATE_NONE || p.phases == 1, "gate+phases unsupported");
    // pick the N tile: wide tiles when the input halo allows it and there are enough columns
    const int halo = (p.KW - 1) * p.dil;
    auto fits = [&](int bn) { return (bn - 1) * p.stride + halo + 1 <= XW_MAX; };
    if (p.CoutP % 128 == 0) {
        if (p.Nout >= 96 && fits(128) && (p.Nout % 128 == 0 || p.Nout % 128 > 64 || p.Nout >= 2048))
            launch_cfg<128, 128, 2, 2>(p, stream, "conv_gemm_kernel<128,128,2,2>");
        else {
            DTTS_REQUIRE(fits(64), "conv halo too large");
            launch_cfg<128, 64, 2, 2>(p, stream, "conv_gemm_kernel<128,64,2,2>");
        }
    } else if (p.CoutP % 64 == 0) {
        if (fits(128) && p.Nout > 64) launch_cfg<64, 128, 2, 2>(p, stream, "conv_gemm_kernel<64,128,2,2>");
        else { DTTS_REQUIRE(fits(64), "conv halo too large"); launch_cfg<64, 64, 2, 2>(p, stream, "conv_gemm_kernel<64,64,2,2>"); }
    } else {
        DTTS_REQUIRE(p.CoutP % 32 == 0, "CoutP must be a multiple of 32");
        DTTS_REQUIRE(fits(128), "conv halo too large");
        launch_cfg<32, 128, 1, 4>(p, stream, "conv_gemm_kernel<32,128,1,4>");
    }
}

}  // namespace dtts
