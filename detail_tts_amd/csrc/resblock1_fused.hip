// LDS-resident fused HiFiGAN ResBlock1 group for the NARROW generator stages (25 and 12 channels at 120 k / 240 k samples):
// the three ResBlock1 branches of one stage (kernels 3 / 7 / 11; vqvae/modules/modules.py:240-334) and their mean
// (vqvae/model_24k.py:276-283:  xs = sum_j resblocks[i * 3 + j](x);  x = xs / 3) in ONE kernel.
//
// Launch by launch these stages are 18 convs of 12 -> 12 / 25 -> 25 channels, each reading and writing the whole activation
// (92 MB per pass at batch 8): 36 passes + the branch mean against the 2 the algorithm needs (read x, write the mean).  Here a
// workgroup owns a TIME TILE of one sample: the tile + a 60-sample halo per side (the receptive field of the widest branch:
// sum over its three layer pairs of (k - 1)(d + 1) / 2 = 6 (k - 1)) is loaded once per branch, all six convs of the branch run out of two
// LDS buffers, and only the mean leaves the CU.  HBM traffic: (1 + 120 / TT) reads x 3 branches (L2-resident after the first)
// + 1 write.
//
// Arithmetic: exact fp32 on v_mfma_f32_16x16x4_f32 (the padded channel counts 16 / 32 are one or two 16-row tiles; K = (4 input
// channels) per instruction, looped over channel blocks and taps).  A conv is
//     acc[16 co, 16 t] += W[16 co, 4 ci] (tap) * src[4 ci, 16 t + tap * dil - pad]
// with the B operand read straight from the activation buffer (ds_read_b32: 4 rows x 16 consecutive floats, row stride = 16 mod 32
// banks -> conflict-free) and the A operand (weights, L2 / L1-resident) loaded one (channel block, tap) ahead.
// The leaky-relu prologue of every conv is applied by its PRODUCER (the buffers hold activated values); the raw residual stream
// and the running mean stay in registers in the MFMA accumulator layout.
// Sequence ends: positions outside [0, len) are zero for every conv input (F.conv1d's zero padding at the true sequence edges),
// so every layer's output is masked; tile edges need nothing: what a truncated halo corrupts never reaches the tile's interior.
#include <algorithm>

#include "ops.h"
#include "prof.h"
#include "split3.h"

namespace dtts {

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RBF_H = 60;        // halo per side: 6 (k - 1) for k = 11
constexpr int RBF_G = 40;        // guard columns per side of a row (>= the largest conv padding, 25), never meaningful

template <int CP>
struct RbfGeo {
    static constexpr int W = CP == 16 ? 1088 : 512;          // staged columns (tile + 2 halos), a multiple of 64
    static constexpr int WP = W + 2 * RBF_G;                 // row stride: 16 mod 32 floats
    static constexpr int TT = W - 2 * RBF_H;                 // interior
    static constexpr int NT = W / 64;                        // 16-column tiles per wave
    static constexpr int MT = CP / 16;
    static_assert(WP % 32 == 16, "row stride must be 16 mod 32 banks");
    static_assert(TT % 4 == 0, "interior must keep 16-byte stores aligned");
};

__device__ __forceinline__ float lrelu01(float v) { return fmaxf(v, 0.1f * v); }

// one conv of the chain: dst = f(conv(src)) on this wave's NT column tiles.
//   MODE 0 (convs1): dst <- lrelu(mask(acc))                          (input of convs2)
//   MODE 1 (convs2): res <- mask(acc + res); dst <- lrelu(res)        (residual stream in registers; input of the next layer)
template <int CP, int K, int MODE>
__device__ __forceinline__ void rbf_conv(const float* __restrict__ w, const float* __restrict__ bias, int CinP, int CoutP, int dil,
                                         const float* src, float* dst, float (&res)[RbfGeo<CP>::NT][RbfGeo<CP>::MT][4], int lane,
                                         int wcol0, int tglob0, int len) {
    using G = RbfGeo<CP>;
    constexpr int NT = G::NT, MT = G::MT, WP = G::WP;
    const int l16 = lane & 15, lq = lane >> 4;
    const int pad = (K - 1) * dil / 2;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = bias ? bias[16 * mt + 4 * lq + r] : 0.f;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[ct][mt] = b4;
    }
    // A operand of (channel block cb, tap): lane (co = 16 mt + l16, ci = 4 cb + lq)
    const float* wl = w + (long long)lq * CoutP + l16;
    const float* sl = src + lq * WP + RBF_G + wcol0 + l16 - pad;
    constexpr int NIT = (CP / 4) * K;
    // both operands of iteration it + 1 are fetched (weights: global, L1 / L2-resident; activations: LDS) under the MFMAs of iteration it:
    // with one wave per SIMD nothing else would cover their latency
    float a_cur[MT], a_nxt[MT], b_cur[NT], b_nxt[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = wl[16 * mt];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) b_cur[ct] = sl[ct * 16];
    int cb = 0, tap = 0;
    for (int it = 0; it < NIT; ++it) {
        int cbn = cb, tapn = tap + 1;
        if (tapn == K) { tapn = 0; ++cbn; }
        if (it + 1 < NIT) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = wl[((long long)tapn * CinP + 4 * cbn) * CoutP + 16 * mt];
            const float* sp = sl + (4 * cbn) * WP + tapn * dil;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) b_nxt[ct] = sp[ct * 16];
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ct][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mt], b_cur[ct], acc[ct][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) b_cur[ct] = b_nxt[ct];
        cb = cbn;
        tap = tapn;
    }
    // epilogue: lane holds rows 16 mt + 4 lq + r of column wcol0 + 16 ct + l16
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int col = wcol0 + ct * 16 + l16, t = tglob0 + col;
        const bool valid = t >= 0 && t < len;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[ct][mt][r];
                if (MODE == 1) v += res[ct][mt][r];
                v = valid ? v : 0.f;
                if (MODE == 1) res[ct][mt][r] = v;
                dst[(16 * mt + 4 * lq + r) * WP + RBF_G + col] = lrelu01(v);
            }
    }
}

template <int CP, int K>
__device__ __forceinline__ void rbf_branch(const RbFusedParams& p, int br, float* bufL, float* bufB,
                                           float (&res)[RbfGeo<CP>::NT][RbfGeo<CP>::MT][4], int lane, int wcol0, int tglob0, int len) {
#pragma unroll 1
    for (int li = 0; li < 3; ++li) {
        const int q = (br * 3 + li) * 2;
        rbf_conv<CP, K, 0>(p.w[q], p.b[q], p.CinP, p.CoutP, p.dil[li], bufL, bufB, res, lane, wcol0, tglob0, len);
        __syncthreads();
        rbf_conv<CP, K, 1>(p.w[q + 1], p.b[q + 1], p.CinP, p.CoutP, 1, bufB, bufL, res, lane, wcol0, tglob0, len);
        __syncthreads();
    }
}

template <int CP>
__global__ __launch_bounds__(256, 1) void resblock1x3_fused_kernel(const RbFusedParams p) {
    using G = RbfGeo<CP>;
    constexpr int W = G::W, WP = G::WP, TT = G::TT, NT = G::NT, MT = G::MT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bufL = lds;                      // activated input of the next convs1
    float* bufB = lds + CP * WP;            // activated convs1 output; at a branch start: the raw tile (for the residual registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = p.lens ? p.lens[b] : p.T;
    if (t0 >= len) return;
    const int tglob0 = t0 - RBF_H;          // global time of staged column 0
    const int wcol0 = wave * (W / 4);
    const float* xb = p.x + (long long)b * p.x_bs;
    // guards + padded channel rows: finite (zero) once; every later write to them is zero again
    for (int i = tid; i < 2 * CP * WP; i += 256) lds[i] = 0.f;
    __syncthreads();

    float res[NT][MT][4], mean[NT][MT][4];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mean[ct][mt][r] = 0.f;

#pragma unroll 1
    for (int br = 0; br < 3; ++br) {
        if (!((p.branch_mask >> br) & 1)) continue;
        // stage the tile: raw -> bufB (source of the residual registers), lrelu -> bufL; zero outside [0, len)
        for (int i = tid; i < p.C * (W / 4); i += 256) {
            const int row = i / (W / 4), c4 = i - row * (W / 4), t = tglob0 + c4 * 4;
            float v[4];
            const float* src = xb + (long long)row * p.x_cs + t;
            if (t >= 0 && t + 3 < len && p.vec_ok) {
                const float4 q = *reinterpret_cast<const float4*>(src);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (t + e >= 0 && t + e < len) ? src[e] : 0.f;
            }
            float* db = bufB + row * WP + RBF_G + c4 * 4;
            float* dl = bufL + row * WP + RBF_G + c4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { db[e] = v[e]; dl[e] = lrelu01(v[e]); }
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) res[ct][mt][r] = bufB[(16 * mt + 4 * lq + r) * WP + RBF_G + wcol0 + ct * 16 + l16];
        __syncthreads();
        if (br == 0) rbf_branch<CP, 3>(p, 0, bufL, bufB, res, lane, wcol0, tglob0, len);
        else if (br == 1) rbf_branch<CP, 7>(p, 1, bufL, bufB, res, lane, wcol0, tglob0, len);
        else rbf_branch<CP, 11>(p, 2, bufL, bufB, res, lane, wcol0, tglob0, len);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mean[ct][mt][r] += res[ct][mt][r];
    }
    // mean -> bufB (accumulator layout) -> coalesced 16-byte stores of the interior
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bufB[(16 * mt + 4 * lq + r) * WP + RBF_G + wcol0 + ct * 16 + l16] = mean[ct][mt][r] * p.scale;
    __syncthreads();
    float* yb = p.y + (long long)b * p.y_bs;
    for (int i = tid; i < p.C * (TT / 4); i += 256) {
        const int row = i / (TT / 4), c4 = i - row * (TT / 4), t = t0 + c4 * 4;
        if (t >= len) continue;
        const float* sp = bufB + row * WP + RBF_G + RBF_H + c4 * 4;
        float* dst = yb + (long long)row * p.y_cs + t;
        if (t + 3 < len && p.vec_ok) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(sp);
        else
            for (int e = 0; e < 4 && t + e < len; ++e) dst[e] = sp[e];
    }
}

// =============================================================================================================================
// Split-precision variant (the default): the same tile, the same dataflow, on v_mfma_f32_32x32x16_f16 with every operand as two
// fp16 planes and three cross products (split3.h, conv_x3.h) - the fp32 instruction above is 16 x slower per flop and was the bound
// of the fp32 form (462 GFLOP in 5.9 ms).  Activations live in the LDS as operand planes [plane][8-channel chunk][column][8 fp16]
// (x XS_SCALE_X, leaky-relu applied by the producer), so a B fragment (8 channels of one column per lane) is one ds_read_b128 per
// plane and a tap is a column offset; a conv's epilogue re-splits its output (each lane holds 4 consecutive channels of a column:
// half a chunk) straight into the other buffer.  Weights are pre-split at bind time into A-fragment order
// [K-step = (tap, 16-channel half)][plane][lane][8 fp16] (x XS_SCALE_W).  M = 32 output rows per instruction: 25 -> 32 (78 % used),
// 12 -> 32 (37 %: still 2.7 x the fp32 form).
typedef float f32x16r __attribute__((ext_vector_type(16)));

template <int CP>
struct RbxGeo {
    static constexpr int W = CP == 16 ? 1024 : 512;          // staged columns, a multiple of 128 (4 waves x 32-column tiles)
    static constexpr int G = 32;                             // guard columns per side (>= the largest conv padding, 25)
    static constexpr int WP = W + 2 * G;
    static constexpr int TT = W - 2 * RBF_H;
    static constexpr int NT = W / 128;                       // 32-column tiles per wave
    static constexpr int C8 = CP / 8;                        // 8-channel chunks
    static constexpr int NH = CP / 16;                       // 16-channel K halves per tap
    static constexpr int RG = CP / 8;                        // 4-row register groups of the accumulator that hold real channels... per half
    static constexpr int BUF = 2 * C8 * WP * 16;             // bytes of one activation buffer (both planes)
    static_assert(TT % 4 == 0, "interior must keep 16-byte stores aligned");
    static_assert(BUF >= W * CP * 4, "the fp32 staging of a branch start must fit one buffer");
};

__global__ __launch_bounds__(256) void rbx_pack_weights_kernel(const float* __restrict__ wp, int KW, int CinP, int CoutP, int NH, uint4* __restrict__ out) {
    // out[(kstep * 2 + plane) * 64 + lane], kstep = tap * NH + h16; lane: co = lane & 31, ci = 16 h16 + 8 (lane >> 5) + e
    const int ks = blockIdx.x, lane = threadIdx.x;
    if (lane >= 64) return;
    const int tap = ks / NH, h16 = ks - tap * NH, co = lane & 31, ci0 = 16 * h16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (co < CoutP && ci0 + e < CinP) ? wp[((long long)tap * CinP + ci0 + e) * CoutP + co] * XS_SCALE_W : 0.f;
    uint4 q0, q1;
    split8(v, q0, q1);
    out[(ks * 2 + 0) * 64 + lane] = q0;
    out[(ks * 2 + 1) * 64 + lane] = q1;
}

template <int CP, int K, int MODE, bool SAT>
__device__ __forceinline__ void rbx_conv(const uint4* __restrict__ wf, const float* __restrict__ bias, int dil, const unsigned char* src,
                                         unsigned char* dst, float (&res)[RbxGeo<CP>::NT][CP / 2], int lane, int wcol0, int tglob0, int len,
                                         int* sat) {
    using Gm = RbxGeo<CP>;
    constexpr int NT = Gm::NT, WP = Gm::WP, C8 = Gm::C8, NH = Gm::NH, NKS = K * NH, NR = CP / 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int pad = (K - 1) * dil / 2;
    f32x16r acc[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    // B fragment of (tap, h16): chunk c8 = 2 h16 + lhi, column wcol0 + 32 ct + l31 + tap dil - pad
    const unsigned char* sl = src + ((long long)lhi * WP + Gm::G + wcol0 + l31 - pad) * 16;
    hf8 a_cur[2], a_nxt[2];
    a_cur[0] = __builtin_bit_cast(hf8, wf[lane]);
    a_cur[1] = __builtin_bit_cast(hf8, wf[64 + lane]);
    int tap = 0, h16 = 0;
    for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) {
            a_nxt[0] = __builtin_bit_cast(hf8, wf[(ks + 1) * 128 + lane]);
            a_nxt[1] = __builtin_bit_cast(hf8, wf[(ks + 1) * 128 + 64 + lane]);
        }
        const unsigned char* sp = sl + ((long long)(2 * h16) * WP + tap * dil) * 16;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const hf8 b0 = *reinterpret_cast<const hf8*>(sp + ct * 512);
            const hf8 b1 = *reinterpret_cast<const hf8*>(sp + ct * 512 + C8 * WP * 16);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[1], b0, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0], b1, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[0], b0, acc[ct], 0, 0, 0);
        }
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
        if (++h16 == NH) { h16 = 0; ++tap; }
    }
    // epilogue: lane (column l31, half lhi) holds rows 8 rg + 4 lhi + (0..3) in registers 4 rg .. 4 rg + 3: half of chunk rg
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int col = wcol0 + ct * 32 + l31, t = tglob0 + col;
        const bool valid = t >= 0 && t < len;
#pragma unroll
        for (int rg = 0; rg < C8; ++rg) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = acc[ct][4 * rg + e] * XS_ACC_SCALE + bias[8 * rg + 4 * lhi + e];
                if (MODE == 1) u += res[ct][4 * rg + e];
                u = valid ? u : 0.f;
                if (MODE == 1) res[ct][4 * rg + e] = u;
                v[e] = fmaxf(u * XS_SCALE_X, u * (0.1f * XS_SCALE_X));      // lrelu(u) * scale (both factors positive)
            }
            if (SAT && !(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) <= 65504.f)) *sat = 1;
            unsigned w0[2], w1[2];
            split_pair(v[0], v[1], w0[0], w1[0]);
            split_pair(v[2], v[3], w0[1], w1[1]);
            unsigned char* o = dst + ((long long)rg * WP + Gm::G + col) * 16 + lhi * 8;
            *reinterpret_cast<uint2*>(o) = make_uint2(w0[0], w0[1]);
            *reinterpret_cast<uint2*>(o + C8 * WP * 16) = make_uint2(w1[0], w1[1]);
        }
    }
}

template <int CP, int K, bool SAT>
__device__ __forceinline__ void rbx_branch(const RbFusedParams& p, int br, unsigned char* bufL, unsigned char* bufB,
                                           float (&res)[RbxGeo<CP>::NT][CP / 2], int lane, int wcol0, int tglob0, int len) {
#pragma unroll 1
    for (int li = 0; li < 3; ++li) {
        const int q = (br * 3 + li) * 2;
        rbx_conv<CP, K, 0, SAT>(static_cast<const uint4*>(p.w3[q]), p.b[q], p.dil[li], bufL, bufB, res, lane, wcol0, tglob0, len, p.sat);
        __syncthreads();
        rbx_conv<CP, K, 1, SAT>(static_cast<const uint4*>(p.w3[q + 1]), p.b[q + 1], 1, bufB, bufL, res, lane, wcol0, tglob0, len, p.sat);
        __syncthreads();
    }
}

template <int CP, bool SAT>
__global__ __launch_bounds__(256, 1) void resblock1x3_fused_x3_kernel(const RbFusedParams p) {
    using Gm = RbxGeo<CP>;
    constexpr int W = Gm::W, WP = Gm::WP, TT = Gm::TT, NT = Gm::NT, C8 = Gm::C8, NR = CP / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* bufL = ldsb;
    unsigned char* bufB = ldsb + Gm::BUF;
    float* stage = reinterpret_cast<float*>(bufB);          // fp32 [CP][W] view of bufB at a branch start / for the final mean
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const int len = p.lens ? p.lens[b] : p.T;
    if (t0 >= len) return;
    const int tglob0 = t0 - RBF_H;
    const int wcol0 = wave * (W / 4);
    const float* xb = p.x + (long long)b * p.x_bs;
    for (int i = tid; i < 2 * Gm::BUF / 16; i += 256) reinterpret_cast<uint4*>(ldsb)[i] = make_uint4(0, 0, 0, 0);      // guards: finite
    __syncthreads();

    float res[NT][NR], mean[NT][NR];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < NR; ++r) mean[ct][r] = 0.f;

#pragma unroll 1
    for (int br = 0; br < 3; ++br) {
        if (!((p.branch_mask >> br) & 1)) continue;
        // stage the tile: thread = (8-channel chunk, column): raw fp32 -> `stage` (source of the residual registers), lrelu + split -> bufL
        for (int i = tid; i < C8 * W; i += 256) {
            const int c8 = i / W, col = i - c8 * W, t = tglob0 + col;
            float v[8], a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 8 * c8 + e;
                v[e] = (ch < p.C && t >= 0 && t < len) ? xb[(long long)ch * p.x_cs + t] : 0.f;
                stage[ch * W + col] = v[e];
                a[e] = lrelu01(v[e]) * XS_SCALE_X;
            }
            if (SAT) {
                float m = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(a[e]));
                if (!(m <= 65504.f)) *p.sat = 1;
            }
            uint4 q0, q1;
            split8(a, q0, q1);
            unsigned char* o = bufL + ((long long)c8 * WP + Gm::G + col) * 16;
            *reinterpret_cast<uint4*>(o) = q0;
            *reinterpret_cast<uint4*>(o + C8 * WP * 16) = q1;
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int rg = 0; rg < C8; ++rg)
#pragma unroll
                for (int e = 0; e < 4; ++e) res[ct][4 * rg + e] = stage[(8 * rg + 4 * lhi + e) * W + wcol0 + ct * 32 + l31];
        __syncthreads();
        // bufB was used as fp32 staging: its guard columns must be zero again before it serves as an operand buffer (every other
        // column is rewritten by the first convs1 epilogue before anything reads it)
        for (int i = tid; i < 2 * C8 * 2 * Gm::G; i += 256) {
            const int row = i / (2 * Gm::G), g = i - row * (2 * Gm::G);
            reinterpret_cast<uint4*>(bufB)[row * WP + (g < Gm::G ? g : W + g)] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        if (br == 0) rbx_branch<CP, 3, SAT>(p, 0, bufL, bufB, res, lane, wcol0, tglob0, len);
        else if (br == 1) rbx_branch<CP, 7, SAT>(p, 1, bufL, bufB, res, lane, wcol0, tglob0, len);
        else rbx_branch<CP, 11, SAT>(p, 2, bufL, bufB, res, lane, wcol0, tglob0, len);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < NR; ++r) mean[ct][r] += res[ct][r];
    }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int rg = 0; rg < C8; ++rg)
#pragma unroll
            for (int e = 0; e < 4; ++e) stage[(8 * rg + 4 * lhi + e) * W + wcol0 + ct * 32 + l31] = mean[ct][4 * rg + e] * p.scale;
    __syncthreads();
    float* yb = p.y + (long long)b * p.y_bs;
    for (int i = tid; i < p.C * (TT / 4); i += 256) {
        const int row = i / (TT / 4), c4 = i - row * (TT / 4), t = t0 + c4 * 4;
        if (t >= len) continue;
        const float* sp = stage + row * W + RBF_H + c4 * 4;
        float* dstp = yb + (long long)row * p.y_cs + t;
        if (t + 3 < len && p.vec_ok) *reinterpret_cast<float4*>(dstp) = *reinterpret_cast<const float4*>(sp);
        else
            for (int e = 0; e < 4 && t + e < len; ++e) dstp[e] = sp[e];
    }
}
}  // namespace

int rb_fused_tile(int CP) { return CP == 16 ? RbfGeo<16>::TT : RbfGeo<32>::TT; }

size_t rb_fused_w3_bytes(int KW, int CP) { return (size_t)KW * (CP / 16) * 2 * 64 * 16; }

void launch_rb_pack_weights(const float* wp, int KW, int CinP, int CoutP, int CP, void* out, hipStream_t s) {
    DTTS_REQUIRE(CP == 16 || CP == 32, "fused ResBlock1 weight packing: padded channels 16 / 32");
    hipLaunchKernelGGL(rbx_pack_weights_kernel, dim3(KW * (CP / 16)), dim3(64), 0, s, wp, KW, CinP, CoutP, CP / 16, static_cast<uint4*>(out));
    DTTS_CHECK_HIP(hipGetLastError());
}

void launch_resblock1x3_fused(const RbFusedParams& p_in, hipStream_t s) {
    RbFusedParams p = p_in;
    DTTS_REQUIRE(p.C >= 1 && p.C <= 32 && p.CinP % 16 == 0 && p.CinP <= 32 && p.CoutP == 32, "fused ResBlock1: 1..32 channels, packed 16 / 32 x 32");
    DTTS_REQUIRE(p.k[0] == 3 && p.k[1] == 7 && p.k[2] == 11, "fused ResBlock1: kernels (3, 7, 11)");
    for (int li = 0; li < 3; ++li) DTTS_REQUIRE((11 - 1) * p.dil[li] / 2 <= RBF_G && p.dil[li] >= 1, "fused ResBlock1: dilation");
    {
        int halo = 0;
        for (int li = 0; li < 3; ++li) halo += (11 - 1) * (p.dil[li] + 1) / 2;
        DTTS_REQUIRE(halo <= RBF_H, "fused ResBlock1: receptive field exceeds the staged halo");
    }
    auto al16 = [](const void* q, long long bs, int cs) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0 && (bs & 3) == 0 && (cs & 3) == 0; };
    p.vec_ok = (al16(p.x, p.x_bs, p.x_cs) && al16(p.y, p.y_bs, p.y_cs)) ? 1 : 0;
    const int CP = p.C <= 16 ? 16 : 32;
    const size_t l16 = sizeof(float) * 2 * 16 * RbfGeo<16>::WP, l32 = sizeof(float) * 2 * 32 * RbfGeo<32>::WP;
    if (!p.w3[0]) lds_optin(CP == 16 ? reinterpret_cast<const void*>(resblock1x3_fused_kernel<16>) : reinterpret_cast<const void*>(resblock1x3_fused_kernel<32>),
                            (int)(CP == 16 ? l16 : l32));
    // FLOPs: 6 convs x (3 + 7 + 11) taps x 2 C^2 per sample (useful channels); bytes: x in + mean out
    int taps = 0;
    for (int j = 0; j < 3; ++j) taps += ((p.branch_mask >> j) & 1) ? p.k[j] : 0;
    const double n = (double)p.B * p.T;
    ProfScope ps("resblock1x3_fused_kernel", 2.0 * 6.0 * taps * p.C * p.C * n, 8.0 * p.C * n, s);
    if (p.w3[0]) {                  // split-precision form (weights pre-split into fragment order)
        for (int q = 0; q < 18; ++q) DTTS_REQUIRE(p.w3[q] && p.b[q], "fused ResBlock1 (split precision): weights / biases");
        const dim3 g16(cdiv(p.T, RbxGeo<16>::TT), p.B), g32(cdiv(p.T, RbxGeo<32>::TT), p.B);
        auto go = [&](auto kern, dim3 g, int lds) {
            lds_optin(reinterpret_cast<const void*>(kern), lds);
            hipLaunchKernelGGL(kern, g, dim3(256), lds, s, p);
        };
        if (CP == 16 && !p.sat) go(resblock1x3_fused_x3_kernel<16, false>, g16, 2 * RbxGeo<16>::BUF);
        else if (CP == 16) go(resblock1x3_fused_x3_kernel<16, true>, g16, 2 * RbxGeo<16>::BUF);
        else if (!p.sat) go(resblock1x3_fused_x3_kernel<32, false>, g32, 2 * RbxGeo<32>::BUF);
        else go(resblock1x3_fused_x3_kernel<32, true>, g32, 2 * RbxGeo<32>::BUF);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (CP == 16) hipLaunchKernelGGL(resblock1x3_fused_kernel<16>, dim3(cdiv(p.T, RbfGeo<16>::TT), p.B), dim3(256), l16, s, p);
    else hipLaunchKernelGGL(resblock1x3_fused_kernel<32>, dim3(cdiv(p.T, RbfGeo<32>::TT), p.B), dim3(256), l32, s, p);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
