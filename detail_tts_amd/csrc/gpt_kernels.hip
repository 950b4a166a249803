// GPT decode-loop kernels for gfx950 (see gpt_kernels.h).  The decode step streams ~308 MB of fp32 weights
// per token regardless of the batch -> these kernels are HBM-bound: wide (16 B/lane) coalesced
// weight loads, many waves in flight, wave64 shuffle reductions, no MFMA.
#include "gpt_kernels.h"
#include <cstdlib>
#include "philox.h"

namespace dtts {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------ LayerNorm on vectors
__device__ __forceinline__ void block_ln_256(float* v, int n_per, int C, const float* g, const float* bta, float* red) {
    // v: this thread's n_per values (strided by 256 over C). Normalises in place.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int i = 0; i < n_per; ++i) s += v[i];
    s = wsum(s);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    float q = 0.f;
    for (int i = 0; i < n_per; ++i) {
        const int c = tid + i * 256;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    q = wsum(q);
    __syncthreads();
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)C + 1e-5f);
    for (int i = 0; i < n_per; ++i) {
        const int c = tid + i * 256;
        if (c < C) v[i] = (v[i] - mean) * rstd * g[c] + bta[c];
    }
}

// ------------------------------------------------------------------------------------------ decode step (graph-capturable)
// Every value that changes from token to token or from call to call lives in the device-side GptCtl block: a decode step is a
// fixed sequence of kernel launches with fixed arguments, so it can be captured once in a hipGraph and replayed.
//
// Per layer FIVE launches (the LayerNorms and the split-K finishes live in the consumers' prologues):
//   K1 c_attn GEMV   in: X = res + bias + sum(partials of the previous c_proj(mlp)) -> writes X and its rows' (sum, sum sq);  W (gamma . X)
//   K2 attention     q/k/v = LN-algebra finish of K1's partials, KV append, softmax(q K) V
//   K3 c_proj GEMV   in: attention output
//   K4 c_fc GEMV     in: Y = X + bias + sum(K3 partials)                             -> writes Y and its statistics;           W (gamma . Y)
//   K5 c_proj(mlp)   in: gelu(LN-algebra finish of K4's partials)
// LN algebra: W^T LN(y) = r (W^T (gamma . y) - mu c) + d with c = W^T gamma, d = W^T beta + bias (precomputed at bind time,
// launch_ln_fold_vectors), r = rstd, mu = mean: the GEMV never needs the row statistics, its CONSUMER applies them as two scalars
// per row.  Only the order of fp32 sums changes; nothing is approximated.

// (mean, rstd) of row b from the per-slice partial sums st[slice][B][2]
__device__ __forceinline__ void ln_row_stats(const float* __restrict__ st, int nsl, int B, int b, int K, float& mean, float& rstd) {
    float S = 0.f, Q = 0.f;
    for (int i = 0; i < nsl; ++i) { S += st[((long long)i * B + b) * 2]; Q += st[((long long)i * B + b) * 2 + 1]; }
    mean = S / (float)K;
    rstd = rsqrtf(fmaxf(Q / (float)K - mean * mean, 0.f) + 1e-5f);
}

// Sum over the split-K slices of element (b, k).  The partials were written by the previous kernel's other CUs, so every load is
// an L2 / fabric round trip: ALL loads of the element are issued before the first add (slice counts are 6, 12 or 24: a switch on
// the uniform count selects a fully unrolled body; a rolled loop would expose one round trip per 4 slices).
template <int N>
__device__ __forceinline__ void load_parts_n(const float* __restrict__ p, long long slice_stride, float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[(long long)i * slice_stride];
}
template <int N>
__device__ __forceinline__ float reduce_parts_n(const float (&v)[N]) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) { a0 += v[i]; a1 += v[i + 1]; a2 += v[i + 2]; a3 += v[i + 3]; }
#pragma unroll
    for (int i = N & ~3; i < N; ++i) a0 += v[i];
    return (a0 + a1) + (a2 + a3);
}
template <int N>
__device__ __forceinline__ float sum_parts_n(const float* __restrict__ p, long long slice_stride) {
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[(long long)i * slice_stride];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i + 4 <= N; i += 4) { a0 += v[i]; a1 += v[i + 1]; a2 += v[i + 2]; a3 += v[i + 3]; }
#pragma unroll
    for (int i = N & ~3; i < N; ++i) a0 += v[i];
    return (a0 + a1) + (a2 + a3);
}
__device__ __forceinline__ float sum_parts(const float* __restrict__ parts, int nsl, long long slice_stride, long long off) {
    const float* p = parts + off;
    switch (nsl) {
        case 1: return p[0];
        case 6: return sum_parts_n<6>(p, slice_stride);
        case 12: return sum_parts_n<12>(p, slice_stride);
        case 16: return sum_parts_n<16>(p, slice_stride);
        case 24: return sum_parts_n<24>(p, slice_stride);
        default: break;
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int sl = 0;
    for (; sl + 4 <= nsl; sl += 4) {
        const float v0 = p[0], v1 = p[slice_stride], v2 = p[2 * slice_stride], v3 = p[3 * slice_stride];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        p += 4 * slice_stride;
    }
    for (; sl < nsl; ++sl) { a0 += p[0]; p += slice_stride; }
    return (a0 + a1) + (a2 + a3);
}

// final LayerNorms of a token: h = res + bias + sum(partials) ; lat = final_norm(ln_f(h)) -> lat[b][C] and latents[b, :, step]
__global__ __launch_bounds__(256) void gpt_final_ln_kernel(const float* res, const float* bias, const float* parts, int in_slices, int in_stride,
                                                           int B, const float* g1, const float* b1, const float* g2, const float* b2,
                                                           float* y, int C, const GptCtl* ctl) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    constexpr int NPER = 4;                  // C <= 1024; every element's loads are issued before the first LayerNorm barrier
    float v[NPER];
    const int n_per = (C + 255) / 256;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int c = tid + i * 256;
        float a = 0.f;
        if (c < C) {
            a = res[(long long)b * C + c] + (bias ? bias[c] : 0.f);
            if (in_slices) a += sum_parts(parts, in_slices, (long long)B * in_stride, (long long)b * in_stride + c);
        }
        v[i] = a;
    }
    block_ln_256(v, n_per, C, g1, b1, red);
    block_ln_256(v, n_per, C, g2, b2, red);
    const int step = ctl->step[b];
    float* col = (ctl->latents && step < ctl->max_steps) ? ctl->latents + (long long)b * ctl->lat_bs + step : nullptr;
#pragma unroll
    for (int i = 0; i < NPER; ++i) {
        const int c = tid + i * 256;
        if (c < C) {
            y[(long long)b * C + c] = v[i];
            if (col) col[(long long)c * ctl->lat_cs] = v[i];
        }
    }
}

void launch_gpt_final_ln(const float* res, const float* bias, const float* parts, int in_slices, int in_stride, int B, const float* g1,
                         const float* b1, const float* g2, const float* b2, float* y, int C, const GptCtl* ctl, hipStream_t s) {
    DTTS_REQUIRE(C <= 1024, "final LayerNorm width");
    hipLaunchKernelGGL(gpt_final_ln_kernel, dim3(B), dim3(256), 0, s, res, bias, parts, in_slices, in_stride, B, g1, b1, g2, b2, y, C, ctl);
    DTTS_CHECK_HIP(hipGetLastError());
}

// c[n] = sum_k gamma[k] W[k][n] ; d[n] = sum_k beta[k] W[k][n] + bias[n]   (bind time; fixed summation order)
__global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const float* __restrict__ W, int K, int CoutP, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              float* __restrict__ c, float* __restrict__ d) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= CoutP) return;
    float sc = 0.f, sd = 0.f;
    for (int k = 0; k < K; ++k) {
        const float w = W[(long long)k * CoutP + n];
        sc += gamma[k] * w;
        sd += beta[k] * w;
    }
    c[n] = sc;
    d[n] = sd + (bias ? bias[n] : 0.f);
}

void launch_ln_fold_vectors(const float* W, int K, int CoutP, const float* gamma, const float* beta, const float* bias, float* c, float* d,
                            hipStream_t s) {
    hipLaunchKernelGGL(ln_fold_vectors_kernel, dim3(cdiv(CoutP, 256)), dim3(256), 0, s, W, K, CoutP, gamma, beta, bias, c, d);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ block GEMV (decode)
// One 512-thread workgroup = 8 waves x RPW weight rows x (64 VEC) columns: every lane issues its weight loads up front (16 KiB per
// wave, >= 768 waves per GEMV -> > 8 MiB in flight, what HBM needs), the 8 waves' partial sums are combined through LDS, and a
// workgroup leaves ONE partial per RB = 8 RPW input rows (split-K slices; the consumer sums them).  Input prologues:
//   GP_PLAIN   x[b][k] as is
//   GP_RESSUM  v = res[b][k] + in_bias[k] + sum_slices parts[sl][b][k]  (the residual stream = the previous GEMV's finish); the
//              column-block-0 workgroups store v to y_out and their rows' (sum, sum sq) to stats_out[slice][b]; GEMV input gamma[k] v
//   GP_LNPARTS v = act(r_b (sum_slices parts[sl][b][k] - mu_b c[k]) + d[k]) with (mu_b, r_b) from stats_in: the LN-algebra finish
//              of the producing GEMV (c_fc) + GELU
//   GP_ATTN    v = the attention output: the key-split partial results (m, l, o) of decode_attention_qkv_kernel combined
template <int PRO, int VEC, int RPW, int NB>
__global__ __launch_bounds__(512) void gemv_block_kernel(const float* __restrict__ W, int K, int CoutP, GemvIn in, int B,
                                                         float* __restrict__ part) {
    constexpr int RB = 8 * RPW;                         // RPW weight rows per wave, RB input rows per workgroup (= one partial slice)
    static_assert(NB == 8 || NB == 16, "batch rows per decode step: 8, or 16 (two requests' stage A decoded as one session)");
    static_assert(RB == 64 || RB == 128, "a row's RB values must be one or two whole waves");
    __shared__ __attribute__((aligned(16))) float xs[NB][RB];
    __shared__ __attribute__((aligned(16))) float red[8][4][64][VEC];
    __shared__ float smr[NB][2];
    __shared__ float sst[NB][2][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = blockIdx.x * (64 * VEC) + lane * VEC;
    const int k0 = blockIdx.y * RB;
    const bool cok = col < CoutP;
    // weights first: 16 independent loads per lane stay in flight across the prologue
    float w[RPW][VEC];
    {
        const float* wp = W + (long long)(k0 + wave * RPW) * CoutP + (cok ? col : 0);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(wp + (long long)r * CoutP);
                w[r][0] = t.x; w[r][VEC > 1 ? 1 : 0] = t.y; w[r][VEC > 2 ? 2 : 0] = t.z; w[r][VEC > 3 ? 3 : 0] = t.w;
            } else {
                w[r][0] = wp[(long long)r * CoutP];
            }
        }
    }
    // ---- prologue.  Every global load of it is issued BEFORE the first wait: VMEM loads return in order, so a wait for any of them
    // also waits for the weight loads above, and each further dependent batch (the split-K partials of the producing GEMV, one batch
    // per input element in round 2) was one more L2 / fabric round trip in front of the FMAs (6.8 -> ~5 us per GEMV).
    constexpr int NE = NB * RB / 512;                           // input elements per thread (a wave never straddles two rows)
    const bool lead = PRO == GP_RESSUM && blockIdx.x == 0;       // this workgroup also publishes the residual rows + statistics
    const long long pstride = (long long)B * in.in_stride;
    int eb[NE], ebb[NE], ei[NE], ek[NE];
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int e = q * 512 + tid;
        eb[q] = e / RB;
        ei[q] = e % RB;
        ebb[q] = eb[q] < B ? eb[q] : B - 1;
        ek[q] = k0 + ei[q];
    }
    float psum[NE], aux0[NE], aux1[NE];                         // summed partials; RESSUM: residual + bias, gamma; LNPARTS: fold c, d
#pragma unroll
    for (int q = 0; q < NE; ++q) psum[q] = aux0[q] = aux1[q] = 0.f;
    if (PRO == GP_RESSUM) {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            aux0[q] = in.x[(long long)ebb[q] * in.x_stride + ek[q]];
            if (in.in_bias) aux0[q] += in.in_bias[ek[q]];
            aux1[q] = in.gamma[ek[q]];
        }
    } else if (PRO == GP_LNPARTS) {
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            aux0[q] = in.fold_c[ek[q]];
            aux1[q] = in.fold_d[ek[q]];
        }
    }
    if (PRO == GP_RESSUM || PRO == GP_LNPARTS) {
        const float* pp[NE];
#pragma unroll
        for (int q = 0; q < NE; ++q) pp[q] = in.parts + (long long)ebb[q] * in.in_stride + ek[q];
        // the loads of ALL elements first, then the sums (each in sum_parts_n's order: bit-identical to summing element by element)
#define DTTS_GEMV_PARTS(N)                                                      \
    {                                                                           \
        float pv[NE][N];                                                        \
        _Pragma("unroll") for (int q = 0; q < NE; ++q) load_parts_n<N>(pp[q], pstride, pv[q]); \
        _Pragma("unroll") for (int q = 0; q < NE; ++q) psum[q] = reduce_parts_n<N>(pv[q]);     \
    }
        switch (in.in_slices) {
            case 0: break;
            case 1: DTTS_GEMV_PARTS(1) break;
            case 6: DTTS_GEMV_PARTS(6) break;
            case 12: DTTS_GEMV_PARTS(12) break;
            case 16: DTTS_GEMV_PARTS(16) break;
            case 24: DTTS_GEMV_PARTS(24) break;
            default:
#pragma unroll
                for (int q = 0; q < NE; ++q) psum[q] = sum_parts(in.parts, in.in_slices, pstride, (long long)ebb[q] * in.in_stride + ek[q]);
        }
#undef DTTS_GEMV_PARTS
    }
    if (PRO == GP_LNPARTS) {
        if (tid < NB) {
            float m = 0.f, r = 0.f;
            if (tid < B) ln_row_stats(in.stats_in, in.stats_slices, B, tid, in.K_ln, m, r);
            smr[tid][0] = m;
            smr[tid][1] = r;
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NE; ++q) {
        const int b = eb[q], i = ei[q], bb = ebb[q], k = ek[q];
        float v;
        if (PRO == GP_LNPARTS) {
            v = act_apply(smr[bb][1] * (psum[q] - smr[bb][0] * aux0[q]) + aux1[q], in.in_act, 0.f);
        } else if (PRO == GP_ATTN) {
            // parts = [B][H][KS][ATT_REC]: per key split (m, l, o[0..D-1]); in_stride = D, in_slices = KS
            const int D = in.in_stride, h = k / D, c = k - h * D, H = K / D, KS = in.in_slices;
            const float* rec = in.parts + ((long long)(bb * H + h) * KS) * ATT_REC;
            float m = -INFINITY;
            for (int j = 0; j < KS; ++j) m = fmaxf(m, rec[j * ATT_REC]);
            float num = 0.f, den = 0.f;
            for (int j = 0; j < KS; ++j) {
                const float wj = __expf(rec[j * ATT_REC] - m);        // an empty split has m = -inf, l = 0, o = 0 -> weight 0
                num += wj * rec[j * ATT_REC + 2 + c];
                den += wj * rec[j * ATT_REC + 1];
            }
            v = num / den;
        } else if (PRO == GP_RESSUM) {
            v = aux0[q];
            if (in.in_slices) v += psum[q];
            if (lead) {
                if (b < B) in.y_out[(long long)b * K + k] = v;
                const float s1 = wsum(v), s2 = wsum(v * v);           // fixed order -> deterministic
                if (lane == 0) { sst[b][RB == 128 ? (wave & 1) : 0][0] = s1; sst[b][RB == 128 ? (wave & 1) : 0][1] = s2; }
            }
            v *= aux1[q];
        } else {
            v = in.x[(long long)bb * in.x_stride + k];
        }
        xs[b][i] = v;
    }
    __syncthreads();
    if (lead && tid < B) {
        float* st = in.stats_out + ((long long)blockIdx.y * B + tid) * 2;
        st[0] = RB == 128 ? sst[tid][0][0] + sst[tid][1][0] : sst[tid][0][0];
        st[1] = RB == 128 ? sst[tid][0][1] + sst[tid][1][1] : sst[tid][0][1];
    }
    float acc[NB][VEC];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[b][c] = 0.f;
#pragma unroll
    for (int q = 0; q < RPW / 4; ++q)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 xv = *reinterpret_cast<const float4*>(&xs[b][wave * RPW + 4 * q]);      // LDS broadcast
            const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[b][c] += w[4 * q + u][c] * xe[u];
        }
    // combine the 8 waves (rows) through LDS, 4 batch rows per round; wave order fixed -> deterministic
#pragma unroll
    for (int h = 0; h < NB / 4; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < VEC; ++c) red[wave][b][lane][c] = acc[h * 4 + b][c];
        __syncthreads();
        if (tid < 256) {
            const int b = tid >> 6, l = tid & 63;
            float t[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) t[c] = red[0][b][l][c];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv)
#pragma unroll
                for (int c = 0; c < VEC; ++c) t[c] += red[wv][b][l][c];
            const int bo = h * 4 + b, cc = blockIdx.x * (64 * VEC) + l * VEC;
            if (bo < B && cc < CoutP) {
                float* o = part + ((long long)blockIdx.y * B + bo) * CoutP + cc;
#pragma unroll
                for (int c = 0; c < VEC; ++c) o[c] = t[c];
            }
        }
    }
}

template <int VEC, int RPW, int NB>
static void gemv_block_launch_nb(int pro, const float* W, int K, int CoutP, const GemvIn& in, int B, float* part, hipStream_t s) {
    const dim3 grid(cdiv(CoutP, 64 * VEC), K / (8 * RPW));
    if (pro == GP_PLAIN) hipLaunchKernelGGL((gemv_block_kernel<GP_PLAIN, VEC, RPW, NB>), grid, dim3(512), 0, s, W, K, CoutP, in, B, part);
    else if (pro == GP_ATTN) hipLaunchKernelGGL((gemv_block_kernel<GP_ATTN, VEC, RPW, NB>), grid, dim3(512), 0, s, W, K, CoutP, in, B, part);
    else if (pro == GP_RESSUM) hipLaunchKernelGGL((gemv_block_kernel<GP_RESSUM, VEC, RPW, NB>), grid, dim3(512), 0, s, W, K, CoutP, in, B, part);
    else hipLaunchKernelGGL((gemv_block_kernel<GP_LNPARTS, VEC, RPW, NB>), grid, dim3(512), 0, s, W, K, CoutP, in, B, part);
}
template <int VEC, int RPW>
static void gemv_block_launch_v(int pro, const float* W, int K, int CoutP, const GemvIn& in, int B, float* part, hipStream_t s) {
    if (B <= 8) gemv_block_launch_nb<VEC, RPW, 8>(pro, W, K, CoutP, in, B, part, s);
    else gemv_block_launch_nb<VEC, RPW, 16>(pro, W, K, CoutP, in, B, part, s);
}

// Workgroup shape per GEMV: enough workgroups to put one on most CUs.  Wide (256-column, 16-byte loads) when that already yields
// >= 192; else narrow 64-column workgroups; and 64 instead of 128 input rows per workgroup when even those are fewer than 128
// (the 768 x 768 attention projection: 12 column groups x 6 slices).
static int gemv_rows(int K, int CoutP) {
    static const int force = []() { const char* v = getenv("DTTS_GEMV_ROWS"); return v ? atoi(v) : 0; }();
    if (force) return force;
    return ((long long)cdiv(CoutP, 64) * (K / 128) < 128 && K % 64 == 0) ? 64 : 128;
}
static bool gemv_wide(int K, int CoutP) {
    static const int force = []() { const char* v = getenv("DTTS_GEMV_VEC"); return v ? atoi(v) : 0; }();
    return force ? force == 4 : (long long)cdiv(CoutP, 256) * (K / 128) >= 192;
}

int gemv_block_slices(int K, int CoutP) { return gemv_wide(K, CoutP) ? K / 128 : K / gemv_rows(K, CoutP); }

void launch_gemv_block(int pro, const float* W, int K, int CoutP, const GemvIn& in, int B, float* part, hipStream_t s) {
    DTTS_REQUIRE(B >= 1 && B <= GEMV_MAXB && K % 128 == 0 && CoutP % 4 == 0, "gemv_block shape");
    const int rows = gemv_rows(K, CoutP);
    DTTS_REQUIRE((long long)(K / rows) * CoutP <= (long long)GEMV_PART_FLOATS, "gemv partial scratch (per batch row)");
    if (gemv_wide(K, CoutP)) gemv_block_launch_v<4, 16>(pro, W, K, CoutP, in, B, part, s);
    else if (rows == 64) gemv_block_launch_v<1, 8>(pro, W, K, CoutP, in, B, part, s);
    else gemv_block_launch_v<1, 16>(pro, W, K, CoutP, in, B, part, s);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ decode attention
// cache layout per (layer, sample): K [C][cap] (channel-major, keys contiguous) then V [cap][C] (token-major).
// One workgroup per (head, sample, key split): a (head, sample) streams 2 * 48 * n floats (~115 KB at n = 300) and one CU pulls
// ~24 GB/s, so the keys are split over KS workgroups (flash-decoding); each leaves (running max, denominator, unnormalised output)
// and the attention projection's GEMV prologue (GP_ATTN) combines them.  c_attn's finish is folded in: every split sums the qkv GEMV
// partials of its head and applies the LayerNorm algebra (row statistics from the producing GEMV); the LAST split appends k and v to
// the cache and owns the new key.  The token position comes from the device-side control block.
// PROJ (grid.z = 1: no key split): the attention output projection (HF c_proj of the attention block, gpt/model.py via GPT2Attention) is
// applied in the same kernel - a (head, row) workgroup multiplies its normalised 48-vector by the head's 48 rows of W_proj, which it
// loaded into registers (144 per thread: 3 columns x 48 rows) BEFORE the attention, and leaves a per-HEAD partial [H][B][C] that the
// next GEMV's residual prologue sums (16 slices).  One launch per layer less: under the diffusion trunk every dependent launch of the
// decode chain costs ~37 us (tools/pipeline_trace.py), whatever its size.
template <int D, bool PROJ>
__global__ __launch_bounds__(256) void decode_attention_qkv_kernel(const float* __restrict__ part, int slices, int B, int CoutP,
                                                                   const float* __restrict__ stats, int stats_slices,
                                                                   const float* __restrict__ fold_c, const float* __restrict__ fold_d,
                                                                   float* cache, long long cache_bs, int cap, const GptCtl* ctl, int H,
                                                                   float* out, const float* __restrict__ wproj, int wpCoutP) {
    extern __shared__ float sc[];            // [keys of this split] scores / probabilities
    float wreg[PROJ ? D : 1][3];
    if (PROJ) {
        const float* wr = wproj + (long long)(blockIdx.x * D) * wpCoutP + threadIdx.x;
#pragma unroll
        for (int c = 0; c < D; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) wreg[c][j] = wr[(long long)c * wpCoutP + 256 * j];
    }
    __shared__ float red[4];
    __shared__ float qkv_s[3][D];
    const int h = blockIdx.x, b = blockIdx.y, ks = blockIdx.z, KS = gridDim.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = H * D;
    const int n = ctl->lp[b] + ctl->step[b];      // keys including the new one, which sits at column n - 1
    const int pos = n - 1, nc = n - 1;            // nc cached keys; the new key / value come from LDS
    const bool last = ks == KS - 1;
    const int per = (nc + KS - 1) / KS;
    const int s_lo = ks * per, s_hi = min(nc, s_lo + per);        // cached keys of this split
    const int nk = max(s_hi - s_lo, 0) + (last ? 1 : 0);          // + the new key
    float* cb = cache + (long long)b * cache_bs;
    if (tid < 3 * D) {
        const int which = tid / D, c = tid - which * D, col = which * C + h * D + c;
        float mean, rstd;
        ln_row_stats(stats, stats_slices, B, b, C, mean, rstd);
        const float a = sum_parts(part, slices, (long long)B * CoutP, (long long)b * CoutP + col);
        const float v = rstd * (a - mean * fold_c[col]) + fold_d[col];
        qkv_s[which][c] = v;
        if (last) {
            if (which == 1) cb[(long long)(h * D + c) * cap + pos] = v;
            else if (which == 2) cb[(long long)C * cap + (long long)pos * C + h * D + c] = v;
        }
    }
    __syncthreads();
    const float* kp = cb + (long long)(h * D) * cap + s_lo;            // K [C][cap]
    const float* vp = cb + (long long)C * cap + (long long)s_lo * C + h * D;      // V [cap][C]
    float q[D];
    const float scale = rsqrtf((float)D);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = qkv_s[0][c] * scale;
    float mx = -INFINITY;
    const int ncl = nk - (last ? 1 : 0);      // cached keys of this split
    for (int s = tid; s < nk; s += 256) {
        float kv[D];
        if (s < ncl) {
#pragma unroll
            for (int c = 0; c < D; ++c) kv[c] = kp[(long long)c * cap + s];      // D independent coalesced loads in flight
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) kv[c] = qkv_s[1][c];
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += 4) { a0 += q[c] * kv[c]; a1 += q[c + 1] * kv[c + 1]; a2 += q[c + 2] * kv[c + 2]; a3 += q[c + 3] * kv[c + 3]; }
        const float a = (a0 + a1) + (a2 + a3);
        sc[s] = a;
        mx = fmaxf(mx, a);
    }
    mx = wmax(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float l = 0.f;
    for (int s = tid; s < nk; s += 256) {
        const float pr = expf(sc[s] - mx);
        sc[s] = pr;
        l += pr;
    }
    l = wsum(l);
    if (lane == 0) red[wave] = l;
    __syncthreads();
    l = red[0] + red[1] + red[2] + red[3];
    // PV: thread (slot = tid / 12, c4 = tid % 12) owns the float4 channel group c4 of the keys s = slot, slot + 21, ... (V rows are
    // token-major: 12 consecutive lanes read one 192-byte row); 8 independent loads per round.  The 21 slots are then combined
    // through LDS in a fixed order.
    constexpr int D4 = D / 4, SLOTS = 256 / D4;          // 12 float4 per row, 21 key slots (252 threads)
    __shared__ float4 pvs[SLOTS][D4];
    {
        const int slot = tid / D4, c4 = tid - slot * D4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < SLOTS) {
            int s = slot;
            for (; s + 7 * SLOTS < ncl; s += 8 * SLOTS) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(vp + (long long)(s + u * SLOTS) * C + c4 * 4);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float pr = sc[s + u * SLOTS];
                    acc.x += pr * v[u].x; acc.y += pr * v[u].y; acc.z += pr * v[u].z; acc.w += pr * v[u].w;
                }
            }
            {   // tail: up to 7 keys, loads issued together
                float4 v[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const int su = s + u * SLOTS;
                    v[u] = su < ncl ? *reinterpret_cast<const float4*>(vp + (long long)su * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const int su = s + u * SLOTS;
                    const float pr = su < ncl ? sc[su] : 0.f;
                    acc.x += pr * v[u].x; acc.y += pr * v[u].y; acc.z += pr * v[u].z; acc.w += pr * v[u].w;
                }
            }
            if (slot == 0 && last) {                          // the key just produced (value row in LDS)
                const float pr = sc[ncl];
                acc.x += pr * qkv_s[2][c4 * 4]; acc.y += pr * qkv_s[2][c4 * 4 + 1]; acc.z += pr * qkv_s[2][c4 * 4 + 2]; acc.w += pr * qkv_s[2][c4 * 4 + 3];
            }
            pvs[slot][c4] = acc;
        }
    }
    __syncthreads();
    if (PROJ) {
        __shared__ float a_s[D];
        if (tid < D) {
            const int c4 = tid >> 2, e = tid & 3;
            float o = 0.f;
#pragma unroll
            for (int q = 0; q < SLOTS; ++q) o += reinterpret_cast<const float*>(&pvs[q][c4])[e];
            a_s[tid] = o / l;                  // one split: the new key is always there, l > 0
        }
        __syncthreads();
        float* ps = out + ((long long)h * B + b) * wpCoutP + tid;          // per-head partial of the projection [H][B][C]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int c = 0; c < D; c += 4) {
                a0 += a_s[c] * wreg[c][j]; a1 += a_s[c + 1] * wreg[c + 1][j]; a2 += a_s[c + 2] * wreg[c + 2][j]; a3 += a_s[c + 3] * wreg[c + 3][j];
            }
            ps[256 * j] = (a0 + a1) + (a2 + a3);
        }
        return;
    }
    float* rec = out + ((long long)(b * H + h) * KS + ks) * ATT_REC;
    if (tid < D) {
        const int c4 = tid >> 2, e = tid & 3;
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) o += reinterpret_cast<const float*>(&pvs[q][c4])[e];
        rec[2 + tid] = o;                      // unnormalised: sum_s exp(score_s - mx) v_s
    }
    if (tid == 0) { rec[0] = mx; rec[1] = l; }           // an empty split leaves (-inf, 0, 0): weight 0 in the combine
}

int decode_attention_splits() {
    static const int n = []() { const char* v = getenv("DTTS_GPT_KSPLIT"); const int k = v ? atoi(v) : 2; return k < 1 ? 1 : (k > 8 ? 8 : k); }();
    return n;
}

void launch_decode_attention_qkv(const float* part, int slices, int CoutP, const float* stats, int stats_slices, const float* fold_c,
                                 const float* fold_d, float* cache, long long cache_bs, int cache_cs, const GptCtl* ctl, int B, int H, int D,
                                 float* out, hipStream_t s, const float* wproj, int wpCoutP) {
    DTTS_REQUIRE(D == 48, "decode attention head dim");
    if (wproj) {             // attention + output projection: one (head, row) workgroup, no key split; out = per-head partials [H][B][wpCoutP]
        DTTS_REQUIRE(wpCoutP == 3 * 256 && H * D <= wpCoutP, "fused attention projection: 768 output columns");
        DTTS_REQUIRE(sizeof(float) * (size_t)cache_cs + 5376 + 256 <= 64 * 1024, "decode attention: KV cache too long for the LDS score buffer");
        hipLaunchKernelGGL((decode_attention_qkv_kernel<48, true>), dim3(H, B, 1), dim3(256), sizeof(float) * cache_cs, s, part, slices, B, CoutP,
                           stats, stats_slices, fold_c, fold_d, cache, cache_bs, cache_cs, ctl, H, out, wproj, wpCoutP);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    // 64 KiB of LDS per workgroup by default: the dynamic score buffer + 5.1 KiB of static arrays (pvs 4 032 B, qkv_s 576 B, red)
    DTTS_REQUIRE(sizeof(float) * (size_t)cache_cs + 5376 <= 64 * 1024, "decode attention: KV cache too long for the LDS score buffer");
    hipLaunchKernelGGL((decode_attention_qkv_kernel<48, false>), dim3(H, B, decode_attention_splits()), dim3(256), sizeof(float) * cache_cs, s, part,
                       slices, B, CoutP, stats, stats_slices, fold_c, fold_d, cache, cache_bs, cache_cs, ctl, H, out, nullptr, 0);
    DTTS_CHECK_HIP(hipGetLastError());
}

// prefill qkv [B, 3C, L] -> cache: K rows copied, V transposed to token-major
__global__ void kv_to_cache_kernel(const float* qkv, long long bs, int cs, const int* lens, int C, float* cache, long long cache_bs,
                                   int cap) {
    const int row = blockIdx.y, b = blockIdx.z;   // row in [0, 2C): K rows then V rows
    const int len = lens[b];
    const float* src = qkv + (long long)b * bs + (long long)(C + row) * cs;
    float* cb = cache + (long long)b * cache_bs;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < len; t += gridDim.x * blockDim.x) {
        if (row < C) cb[(long long)row * cap + t] = src[t];
        else cb[(long long)C * cap + (long long)t * C + (row - C)] = src[t];
    }
}

void launch_kv_to_cache(const float* qkv, long long bs, int cs, const int* lens, int L, int B, int C, float* cache, long long cache_bs,
                        int cache_cs, hipStream_t s) {
    hipLaunchKernelGGL(kv_to_cache_kernel, dim3(cdiv(L, 128), 2 * C, B), dim3(128), 0, s, qkv, bs, cs, lens, C, cache, cache_bs, cache_cs);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ embeddings
__global__ void build_prefix_kernel(const float* cond, const int* text_ids, int text_stride, const int* text_lens, const float* text_emb,
                                    const float* text_pos, const float* mel_emb, const float* mel_pos, const int* mel_ids,
                                    int mel_stride, const int* mel_lens, int C, int Lmax, float* emb) {
    const int j = blockIdx.x, b = blockIdx.y;          // column
    const int tl = text_lens[b];                        // number of text positions (start + ids + stop)
    const int ml = mel_lens[b];                         // number of mel input tokens (start + history)
    if (j >= 1 + tl + ml) return;
    float* col = emb + (long long)b * C * Lmax + j;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v;
        if (j == 0) v = cond[(long long)b * C + c];
        else if (j <= tl) {
            const int id = text_ids[(long long)b * text_stride + (j - 1)];
            v = text_emb[(long long)id * C + c] + text_pos[(long long)(j - 1) * C + c];
        } else {
            const int k = j - 1 - tl;
            const int id = mel_ids[(long long)b * mel_stride + k];
            v = mel_emb[(long long)id * C + c] + mel_pos[(long long)k * C + c];
        }
        col[(long long)c * Lmax] = v;
    }
}

void launch_build_prefix(const float* cond, const int* text_ids, int text_stride, const int* text_lens, const float* text_emb,
                         const float* text_pos, const float* mel_emb, const float* mel_pos, const int* mel_ids, int mel_stride,
                         const int* mel_lens, int B, int C, int Lmax, float* emb, hipStream_t s) {
    hipLaunchKernelGGL(build_prefix_kernel, dim3(Lmax, B), dim3(256), 0, s, cond, text_ids, text_stride, text_lens, text_emb, text_pos,
                       mel_emb, mel_pos, mel_ids, mel_stride, mel_lens, C, Lmax, emb);
    DTTS_CHECK_HIP(hipGetLastError());
}

__global__ void gather_last_kernel(const float* x, long long bs, int cs, const int* lens, int col_off, int C, float* y) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    y[(long long)b * C + c] = x[(long long)b * bs + (long long)c * cs + lens[b] - 1 + col_off];
}

void launch_gather_last(const float* x, long long bs, int cs, const int* lens, int col_off, int B, int C, float* y, hipStream_t s) {
    hipLaunchKernelGGL(gather_last_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, s, x, bs, cs, lens, col_off, C, y);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ sampler
// HF GenerationMixin._sample logits processing + inverse-CDF multinomial of the Philox spec.
constexpr int SAMP_THREADS = 1024;
constexpr int SORT_MAX = 16384;
static_assert(SAMP_THREADS == 4 * 256, "sampler: one thread per bin of the four radix histograms");

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__device__ float block_reduce_max(float v, float* red) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < SAMP_THREADS / 64; ++i) r = fmaxf(r, red[i]);
    return r;
}
__device__ float block_reduce_sum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < SAMP_THREADS / 64; ++i) r += red[i];
    return r;
}
// exclusive prefix of one value per thread across the block; returns prefix, *total gets the block total
__device__ float block_exclusive_scan(float v, float* red, float* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    __syncthreads();
    float base = 0.f, tot = 0.f;
    for (int i = 0; i < SAMP_THREADS / 64; ++i) {
        if (i < wave) base += red[i];
        tot += red[i];
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SAMP_THREADS) void sampler_kernel(const SamplerParams p) {
    extern __shared__ float smem[];
    float* sv = smem;                                   // [Vpad] processed scores
    const int Vpad = (p.V + 3) & ~3;
    float* skey = sv + Vpad;                            // [SORT_MAX]
    unsigned short* sidx = reinterpret_cast<unsigned short*>(skey + SORT_MAX);   // [SORT_MAX]
    __shared__ float red[SAMP_THREADS / 64];
    __shared__ unsigned hist4[4 * 256];                 // one histogram per radix pass, zeroed once: no barrier pair to recycle one
    __shared__ unsigned sh_u[4];
    __shared__ int sh_i[4];

    const int tid = threadIdx.x, b = blockIdx.x, V = p.V;
#define SSTAMP(k)                                                       \
    do {                                                                \
        if (p.trace && tid == 0 && b == 0) p.trace[k] = wall_clock64(); \
    } while (0)
    SSTAMP(0);
    GptCtl* ctl = p.ctl;
    // Everything the kernel reads from memory that does not depend on the control block is requested FIRST: the token kernel's plain
    // logits row (slices = 1, no bias: one load per score) and the seen flags go to registers while the control block's fields (a
    // chain of dependent scalar loads) arrive; so do the fields the LAST phases need (the Philox stream, the finished flag, the next
    // position embedding), which used to cost one memory round trip each behind a barrier.  Same values, same arithmetic.
    constexpr int PF = 9;
    const bool pf_on = p.slices == 1 && !p.bias && V <= PF * SAMP_THREADS;
    float pf_x[PF];
    unsigned char pf_seen[PF];
    if (pf_on) {
        const float* row = p.parts + (long long)b * p.Vs;
        const unsigned char* seen0 = p.seen + (long long)b * V;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int v = tid + i * SAMP_THREADS;
            pf_x[i] = v < V ? row[v] : 0.f;
            pf_seen[i] = v < V ? seen0[v] : (unsigned char)0;
        }
    }
    const int step = ctl->step[b];
    const float* const forced_u = ctl->forced_u;
    const int u_stride = ctl->u_stride;
    const unsigned long long seed_b = ctl->seed[b];
    const int sample_id_b = ctl->sample_id[b];
    const int fin_in = p.finished[b];
    if (step >= ctl->max_steps) return;                  // a replayed graph may run past the requested length: no-op
    float pos_next = 0.f;                                 // mel_pos[step + 1][tid]: C <= SAMP_THREADS is the common case
    const bool pos_pf = p.x_next && p.C <= SAMP_THREADS;
    if (pos_pf && tid < p.C) pos_next = p.mel_pos[(long long)(step + 1) * p.C + tid];
    const int top_k = ctl->top_k;
    const float top_p = ctl->top_p;
    int token;
    // forced token of this (row, step), or -1: sample.  A row may be forced for a prefix only (UnifiedVoice.inference_speech_tortoise's
    // input_tokens, gpt/model.py:533-537) - the choice is uniform over the workgroup (one row per workgroup)
    const int forced = ctl->forced_tokens ? ctl->forced_tokens[(long long)b * ctl->f_stride + step] : -1;
    if (forced >= 0) {
        token = forced;
    } else {
        // 1. logits = head GEMV partials + bias (its finish), repetition penalty over every id in the row's input_ids, temperature
        const unsigned char* seen = p.seen + (long long)b * V;
        const float rp = ctl->repetition_penalty, temp = ctl->temperature;
        const int eos_off = ctl->suppress_eos ? p.eos : -1;
        const bool typical = ctl->typical_mass > 0.f && ctl->typical_mass < 1.f;
        // (the maximum of the processed scores rides along: top-k keeps it, so it is the softmax maximum of step 3 unless the typical
        // warper - which may drop the most probable token - runs; a maximum does not depend on the order it is taken in)
        float tmax = -INFINITY;
        if (pf_on) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int v = tid + i * SAMP_THREADS;
                if (v < V) {
                    float x = 0.f;
                    x += pf_x[i];
                    if (v == eos_off) x = -INFINITY;
                    if (pf_seen[i]) x = x < 0.f ? x * rp : x / rp;
                    const float y = typical ? x : x / temp;
                    sv[v] = y;
                    tmax = fmaxf(tmax, y);
                }
            }
        } else
#pragma unroll 4
        for (int v = tid; v < V; v += SAMP_THREADS) {
            float x = p.bias ? p.bias[v] : 0.f;
            x += sum_parts(p.parts, p.slices, (long long)p.B * p.Vs, (long long)b * p.Vs + v);
            if (v == eos_off) x = -INFINITY;
            if (seen[v]) x = x < 0.f ? x * rp : x / rp;
            const float y = typical ? x : x / temp;
            sv[v] = y;
            tmax = fmaxf(tmax, y);
        }
        tmax = wmax(tmax);
        if ((tid & 63) == 0) red[tid >> 6] = tmax;
        hist4[tid] = 0;                                     // SAMP_THREADS == 4 * 256
        if (tid == 0) sh_i[1] = 0;
        __syncthreads();
        float mx_early = -INFINITY;
        for (int i = 0; i < SAMP_THREADS / 64; ++i) mx_early = fmaxf(mx_early, red[i]);
        SSTAMP(1);
        // 1b. HF TypicalLogitsWarper (inference_speech_tortoise(typical_sampling=True), gpt/model.py:539; it sits between the repetition
        // penalty and the temperature): key = | -log p - H |, ascending sort of the whole vocabulary by it, keep the keys up to the
        // first whose cumulative probability reaches the mass.  Off the infer path: the full 16 K-slot bitonic sort is fine here.
        if (typical) {
            float m0 = -INFINITY;
            for (int v = tid; v < V; v += SAMP_THREADS) m0 = fmaxf(m0, sv[v]);
            m0 = block_reduce_max(m0, red);
            float z0 = 0.f;
            for (int v = tid; v < V; v += SAMP_THREADS) z0 += expf(sv[v] - m0);
            z0 = block_reduce_sum(z0, red);
            const float lse = m0 + logf(z0);
            float ent = 0.f;
            for (int v = tid; v < V; v += SAMP_THREADS) {
                const float lp = sv[v] - lse;
                if (lp > -INFINITY) ent -= expf(lp) * lp;                   // nansum: p = 0 terms drop out
            }
            ent = block_reduce_sum(ent, red);
            int n2 = 64;
            while (n2 < V) n2 <<= 1;
            for (int i = tid; i < n2; i += SAMP_THREADS) {
                skey[i] = i < V ? fabsf(-(sv[i] - lse) - ent) : INFINITY;
                sidx[i] = i < V ? (unsigned short)i : 0xffff;
            }
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2; i += SAMP_THREADS) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const bool asc = (i & k) == 0;
                            const float a = skey[i], c = skey[ixj];
                            const unsigned short ia = sidx[i], ic = sidx[ixj];
                            const bool gt = (a > c) || (a == c && ia > ic);      // total order: key, then id
                            if (gt == asc) { skey[i] = c; skey[ixj] = a; sidx[i] = ic; sidx[ixj] = ia; }
                        }
                    }
                    __syncthreads();
                }
            // cumulative probability in key order; last = number of positions whose inclusive sum stays below the mass
            const int per = n2 / SAMP_THREADS > 0 ? n2 / SAMP_THREADS : 1, i0 = tid * per;
            float loc = 0.f;
            for (int i = i0; i < i0 + per && i < V; ++i) loc += expf(sv[sidx[i]] - lse);
            float tot;
            float run = block_exclusive_scan(loc, red, &tot);
            int cnt = 0;
            for (int i = i0; i < i0 + per && i < V; ++i) {
                run += expf(sv[sidx[i]] - lse);
                cnt += run < ctl->typical_mass ? 1 : 0;
            }
            if (tid == 0) sh_i[0] = 0;
            __syncthreads();
            if (cnt) atomicAdd(&sh_i[0], cnt);
            __syncthreads();
            const int last = sh_i[0] < V - 1 ? sh_i[0] : V - 1;
            const float thr = skey[last];
            __syncthreads();
            for (int v = tid; v < V; v += SAMP_THREADS) {
                const float key = fabsf(-(sv[v] - lse) - ent);
                sv[v] = key > thr ? -INFINITY : sv[v] / temp;
            }
            __syncthreads();
        }
        // 2. top-k: threshold = k-th largest value (radix select on order-preserving keys)
        if (top_k > 0 && top_k < V) {
            unsigned prefix = 0, mask = 0;
            int krem = top_k;
            for (int pass = 3; pass >= 0; --pass) {
                unsigned* hist = hist4 + pass * 256;
                const int sh = pass * 8;
                // a thread's consecutive candidates in one bin are added at once (the top byte - sign and exponent - puts most of
                // the vocabulary into two or three bins, and 64 lanes adding to one LDS address are served one after the other)
                unsigned cur = 0, cnt = 0;
                for (int v = tid; v < V; v += SAMP_THREADS) {
                    const unsigned k = f2ord(sv[v]);
                    if ((k & mask) == prefix) {
                        const unsigned bin = (k >> sh) & 255u;
                        if (cnt && bin != cur) { atomicAdd(&hist[cur], cnt); cnt = 0; }
                        cur = bin;
                        ++cnt;
                    }
                }
                if (cnt) atomicAdd(&hist[cur], cnt);
                __syncthreads();
                {
                    // EVERY wave walks the histogram itself (no result to publish, no barrier pair): lane l owns bins 255-4l .. 252-4l
                    // (descending); exclusive prefix of the counts above each lane's bins, then the lane whose 4 bins cross `krem`
                    // walks them.  Exactly the serial top-down scan, in 6 shuffle steps.
                    const int lane = tid & 63;
                    const int b0 = 255 - 4 * lane;
                    const int c0 = (int)hist[b0], c1 = (int)hist[b0 - 1], c2 = (int)hist[b0 - 2], c3 = (int)hist[b0 - 3];
                    const int mine = c0 + c1 + c2 + c3;
                    int incl = mine;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int v = __shfl_up(incl, o);
                        if (lane >= o) incl += v;
                    }
                    const int excl = incl - mine;
                    const bool crosses = excl < krem && incl >= krem;
                    int bin_l = 0, rem_l = 0;
                    if (crosses) {
                        int acc = excl, bin = b0;
                        if (acc + c0 >= krem) bin = b0;
                        else if (acc + c0 + c1 >= krem) { acc += c0; bin = b0 - 1; }
                        else if (acc + c0 + c1 + c2 >= krem) { acc += c0 + c1; bin = b0 - 2; }
                        else { acc += c0 + c1 + c2; bin = b0 - 3; }
                        bin_l = bin;
                        rem_l = krem - acc;
                    }
                    const unsigned long long who = __ballot(crosses);
                    const int total = __shfl(incl, 63), h0 = __shfl(c3, 63);
                    int bin, rem;
                    if (who) {
                        const int src = __ffsll((long long)who) - 1;
                        bin = __shfl(bin_l, src);
                        rem = __shfl(rem_l, src);
                    } else {
                        // no lane crosses when the total is < krem: the serial scan then ends at bin 0 with acc = everything above it
                        bin = 0;
                        rem = krem - (total - h0);
                    }
                    prefix |= (unsigned)bin << sh;
                    mask |= 255u << sh;
                    krem = rem;
                }
                SSTAMP(5 - pass);
            }
            const float thr = ord2f(prefix);
            for (int v = tid; v < V; v += SAMP_THREADS)
                if (sv[v] < thr) sv[v] = -INFINITY;
            __syncthreads();
            SSTAMP(6);
        }
        // 3. softmax statistics of the kept set
        float mx = mx_early;
        if (typical) {
            mx = -INFINITY;
            for (int v = tid; v < V; v += SAMP_THREADS) mx = fmaxf(mx, sv[v]);
            mx = block_reduce_max(mx, red);
        }
        SSTAMP(7);
        // 4. top-p (nucleus): ascending sort of the kept candidates, drop the tail whose cumulative prob <= 1 - top_p
        if (top_p < 1.0f) {
            // (the slot counter sh_i[1] was cleared in front of the first barrier)
            for (int v = tid; v < V; v += SAMP_THREADS)
                if (sv[v] > -INFINITY) {
                    const int slot = atomicAdd(&sh_i[1], 1);
                    skey[slot] = sv[v];
                    sidx[slot] = (unsigned short)v;
                }
            __syncthreads();
            const int M = sh_i[1];
            int n2 = 64;
            while (n2 < M) n2 <<= 1;
            const float cut = 1.0f - top_p;
            if (n2 == 64) {
                // top-k 50 leaves <= 64 candidates: ONE WAVE pads, sorts, sums and cuts them with no workgroup barrier in between (21
                // for the sort alone otherwise, 6 more for Z / the scan / the cut).  LDS operations of a wave complete in issue order;
                // the wave barrier only pins the compiler's ordering.  The float operations are those of the general path below, in its
                // order: Z = block_reduce_sum of one term per lane of wave 0 (the other waves add + 0.f), the scan = block_exclusive_scan
                // with one element per thread (`base` = 0.f for wave 0).
                if (tid < 64) {
                    if (tid >= M) { skey[tid] = INFINITY; sidx[tid] = 0xffff; }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    for (int k = 2; k <= 64; k <<= 1)
                        for (int j = k >> 1; j > 0; j >>= 1) {
                            const int i = tid, ixj = i ^ j;
                            const float a = skey[i], c = skey[ixj];
                            const unsigned short ia = sidx[i], ic = sidx[ixj];
                            __builtin_amdgcn_wave_barrier();
                            if (ixj > i) {
                                const bool asc = (i & k) == 0;
                                const bool gt = (a > c) || (a == c && ia > ic);      // total order: value, then id
                                if (gt == asc) { skey[i] = c; skey[ixj] = a; sidx[i] = ic; sidx[ixj] = ia; }
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_wave_barrier();
                        }
                    float z = 0.f;
                    if (tid < M) z += expf(skey[tid] - mx);
                    z = 0.f + wsum(z);
                    float loc = 0.f;
                    if (tid < M) loc += expf(skey[tid] - mx) / z;
                    float inc = loc;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const float t = __shfl_up(inc, o);
                        if (tid >= o) inc += t;
                    }
                    float run = 0.f + inc - loc;
                    if (tid < M) {
                        run += expf(skey[tid] - mx) / z;
                        if (run <= cut && tid != M - 1) sv[sidx[tid]] = -INFINITY;
                    }
                }
                __syncthreads();
                SSTAMP(11);
            } else {
            for (int i = M + tid; i < n2; i += SAMP_THREADS) { skey[i] = INFINITY; sidx[i] = 0xffff; }
            __syncthreads();
            SSTAMP(8);
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int i = tid; i < n2; i += SAMP_THREADS) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const bool asc = (i & k) == 0;
                            const float a = skey[i], c = skey[ixj];
                            const unsigned short ia = sidx[i], ic = sidx[ixj];
                            // total order: value, then id (deterministic for ties)
                            const bool gt = (a > c) || (a == c && ia > ic);
                            if (gt == asc) { skey[i] = c; skey[ixj] = a; sidx[i] = ic; sidx[ixj] = ia; }
                        }
                    }
                    __syncthreads();
                }
            SSTAMP(9);
            // Z over the kept set, then inclusive cumulative prob in ascending order
            float z = 0.f;
            for (int i = tid; i < M; i += SAMP_THREADS) z += expf(skey[i] - mx);
            z = block_reduce_sum(z, red);
            SSTAMP(10);
            const int per = (n2 + SAMP_THREADS - 1) / SAMP_THREADS;
            const int i0 = tid * per;
            float loc = 0.f;
            for (int i = i0; i < i0 + per && i < M; ++i) loc += expf(skey[i] - mx) / z;
            float tot;
            float run = block_exclusive_scan(loc, red, &tot);
            for (int i = i0; i < i0 + per && i < M; ++i) {
                run += expf(skey[i] - mx) / z;
                if (run <= cut && i != M - 1) sv[sidx[i]] = -INFINITY;
            }
            __syncthreads();
            SSTAMP(11);
            }
        }
        // 5. inverse-CDF draw in vocabulary order
        const int per = (V + SAMP_THREADS - 1) / SAMP_THREADS;
        const int v0 = tid * per;
        float loc = 0.f;
        for (int v = v0; v < v0 + per && v < V; ++v) loc += (sv[v] > -INFINITY) ? expf(sv[v] - mx) : 0.f;
        float tot;
        float run = block_exclusive_scan(loc, red, &tot);
        SSTAMP(12);
        float u;
        if (forced_u) u = forced_u[(long long)b * u_stride + step];
        else {
            float uu[4];
            philox_uniform4(seed_b, (unsigned)sample_id_b, STAGE_GPT_SAMPLE, step, 0u, uu);
            u = uu[0];
        }
        const float target = u * tot;
        if (tid == 0) { sh_i[2] = V; sh_i[3] = -1; }
        __syncthreads();
        int last_kept = -1;
        for (int v = v0; v < v0 + per && v < V; ++v) {
            if (sv[v] > -INFINITY) {
                run += expf(sv[v] - mx);
                last_kept = v;
                if (run > target) { atomicMin(&sh_i[2], v); break; }
            }
        }
        if (last_kept >= 0) atomicMax(&sh_i[3], last_kept);
        __syncthreads();
        token = sh_i[2] < V ? sh_i[2] : sh_i[3];
        SSTAMP(13);
    }
    const bool fin = fin_in != 0;
    if (fin) token = p.eos;
    __syncthreads();
    if (tid == 0) {
        p.codes[(long long)b * p.codes_stride + step] = token;
        p.seen[(long long)b * V + token] = 1;
        if (token == p.eos) p.finished[b] = 1;
        ctl->step[b] = step + 1;                 // only this workgroup reads or writes step[b] inside this launch
    }
    // next input embedding: mel_embedding[token] + mel_pos_embedding[step + 1]   (gpt/model.py:134-136 with position k)
    if (pos_pf) {
        if (tid < p.C) p.x_next[(long long)b * p.C + tid] = p.mel_emb[(long long)token * p.C + tid] + pos_next;
    } else if (p.x_next)
        for (int c = tid; c < p.C; c += SAMP_THREADS)
            p.x_next[(long long)b * p.C + c] = p.mel_emb[(long long)token * p.C + c] + p.mel_pos[(long long)(step + 1) * p.C + c];
    SSTAMP(14);
#undef SSTAMP
}

void launch_sampler(const SamplerParams& p, hipStream_t s) {
    DTTS_REQUIRE(p.V <= SORT_MAX && p.V < 65535, "vocabulary too large for the LDS sampler");
    const size_t lds = sizeof(float) * (size_t)(((p.V + 3) & ~3) + SORT_MAX) + sizeof(unsigned short) * SORT_MAX;
    lds_optin(reinterpret_cast<const void*>(sampler_kernel), 150 * 1024);
    // DTTS_SAMPLER_TRACE = n: the n-th launch records wall-clock stamps of row 0's phases and prints them (100 MHz clock)
    static const int trace_at = []() { const char* v = getenv("DTTS_SAMPLER_TRACE"); return v ? atoi(v) : 0; }();
    static int launches = 0;
    static long long* d_trace = nullptr;
    SamplerParams q = p;
    q.trace = nullptr;
    const bool tracing = trace_at > 0 && ++launches == trace_at;
    if (tracing) {
        if (!d_trace) DTTS_CHECK_HIP(hipMalloc(&d_trace, sizeof(long long) * 16));
        DTTS_CHECK_HIP(hipMemsetAsync(d_trace, 0, sizeof(long long) * 16, s));
        q.trace = d_trace;
    }
    hipLaunchKernelGGL(sampler_kernel, dim3(p.B), dim3(SAMP_THREADS), lds, s, q);
    DTTS_CHECK_HIP(hipGetLastError());
    if (tracing) {
        long long h[16];
        DTTS_CHECK_HIP(hipMemcpyAsync(h, d_trace, sizeof(h), hipMemcpyDeviceToHost, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));
        static const char* names[15] = {"start", "logits", "radix3", "radix2", "radix1", "radix0", "top-k cut", "max", "compact", "sort", "Z", "top-p", "cdf scan", "draw", "end"};
        fprintf(stderr, "[sampler trace] row 0, us since the kernel's first instruction:");
        for (int k = 1; k < 15; ++k) fprintf(stderr, " %s %.2f |", names[k], h[k] ? (h[k] - h[0]) * 0.01 : -1.0);
        fprintf(stderr, "\n");
    }
}

}  // namespace dtts
