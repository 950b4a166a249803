// Small HBM-bound kernels: normalisation statistics, elementwise glue, sampler updates, RNG.
#pragma once
#include "common.h"

namespace dtts {

// GroupNorm statistics -> per (b, c) affine coefficients (a, d) so that GN(x)[c] == a*x + d,
// optionally composed with the AdaGN scale/shift of the diffusion ResBlock
// (vqvae/diff_model.py:111-115):  a = rstd*gamma*(1+scale), d = (beta - mean*rstd*gamma)*(1+scale) + shift.
// ada (or null): scale for channel c at ada[c*ada_stride], shift at ada[(C+c)*ada_stride] (shared by the batch),
// or per-sample when ada_bs != 0 (ada + b*ada_bs), or at ada + ada_idx[b] when a per-sample index array is given (a batch whose
// samples sit at different sampling steps).
void launch_gn_coeffs(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int groups,
                      const float* gamma, const float* beta, float eps, const float* ada, int ada_stride, int ada_bs,
                      float* ab_out, hipStream_t s, const int* ada_idx = nullptr);

// y[b,c,t*up + u] = act(a*x[b,c,t] + d)   (GroupNorm apply, optional nearest-neighbour upsample by `up`)
void launch_affine_apply(const float* x, long long x_bs, int x_cs, const float* ab, const int* lens, int T, int B, int C,
                         int up, int act, float* y, long long y_bs, int y_cs, hipStream_t s);

// LayerNorm over the channel axis of [B,C,T] (vqvae/modules/modules.py:36-48), y = LN(x + r) (r may be null);
// columns t >= len are left untouched.
void set_ln_channels_reg(bool on);     // 1 (default): norms of <= 1024 channels keep a thread's channels in registers (one load pass); bit-identical
void launch_ln_channels(const float* x, const float* r, long long bs, int cs, const int* lens, int T, int B, int C,
                        const float* gamma, const float* beta, float eps, float* y, long long y_bs, int y_cs, hipStream_t s);

// out[b,c] = mean_t x[b,c,t] over t < len
void launch_mean_time(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, float* out, hipStream_t s);

// y[b,c,t] = v[c] for t < T (broadcast a channel vector)
void launch_broadcast_channels(const float* v, int B, int C, int T, float* y, long long y_bs, int y_cs, hipStream_t s);

// sinusoidal timestep embedding table, vqvae/diff_model.py:20-38: out[c, i] for timestep ts[i]; cos half first. out is [dim, n]
void launch_timestep_sinusoid(const int* ts, int n, int dim, float* out, hipStream_t s);

// Philox normal fill (spec: oracle/philox.py / philox.h): out[b, 0..n) for (seed, sample_ids[b], stage, step), scaled
void launch_philox_normal(float* out, long long bs, int n, int B, unsigned long long seed, const int* sample_ids, int stage,
                          int step, float scale, hipStream_t s);

struct DiffStepCoefs {   // fp32 casts of the float64 tables (vqvae/utils/diffusion.py:1315)
    float sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, min_log, max_log, cfk;
    int nonzero;
};
// One ancestral sampler update (vqvae/utils/diffusion.py:311-386, 480-485).  model_out is the [2B, 2C, T] output of
// the batched (cond | uncond) forward; x [B,C,T] is updated in place; noise is generated from the Philox spec.
// When `final_denorm` the result is mapped through denormalize_torch_mel (vqvae/model_24k.py:508-509).
void launch_diff_update(float* x, long long x_bs, int x_cs, const float* model_out, long long m_bs, int m_cs, const int* lens,
                        int T, int B, int C, DiffStepCoefs k, unsigned long long seed, const int* sample_ids, int step,
                        const float* noise_override, int final_denorm, hipStream_t s, float* x0_out = nullptr);   // x0_out: pred_xstart [B,C,T]

// y = a*x + b*z (generic elementwise with optional exp on second operand) used by the flow prior:
// z_p = m + noise * exp(logs) * noise_scale  (vqvae/model_24k.py:860)
void launch_flow_prior(const float* stats, long long s_bs, int s_cs, const int* lens, int T, int B, int C, float noise_scale,
                       unsigned long long seed, const int* sample_ids, const float* noise_override, int flip, float* z,
                       long long z_bs, int z_cs, hipStream_t s);

// x1 <- (x1 - m) on channels [c0, c0+C) then channel flip of the whole [B, Ctot, T] tensor fused:
// coupling reverse step + Flip (vqvae/modules/modules.py:393-400, 471-475).
void launch_coupling_reverse(const float* x, const float* m, float* y, long long bs, int cs, const int* lens, int T, int B,
                             int Ctot, int flip, hipStream_t s);

// y[b,c,t] = alpha * (x0 + x1 + x2)  — mean of the three ResBlock1 branches (vqvae/model_24k.py:277-283)
// LDS-resident fused ResBlock1 group of a narrow generator stage (resblock1_fused.hip): y = scale * sum over the branches in
// `branch_mask` of ResBlock1_j(x), kernels (3, 7, 11), three (convs1 dilation dil[l], convs2 dilation 1) layer pairs each.
// w / b: packed fp32 weights [KW][CinP][CoutP] / biases [CoutP] of conv (branch j, layer l, which) at index (j * 3 + l) * 2 + which.
struct RbFusedParams {
    const float* x = nullptr;
    float* y = nullptr;
    long long x_bs = 0, y_bs = 0;
    int x_cs = 0, y_cs = 0;
    const int* lens = nullptr;     // device, per sample (null: T)
    int B = 0, C = 0, T = 0, CinP = 0, CoutP = 0;
    int k[3] = {3, 7, 11}, dil[3] = {1, 3, 5};
    const float* w[18] = {};
    const float* b[18] = {};
    const void* w3[18] = {};       // split-precision weights in A-fragment order (launch_rb_pack_weights); all set -> the fp16-pipe kernel
    int branch_mask = 7;
    float scale = 1.f / 3.f;
    int vec_ok = 0;                // set by the launcher
    int* sat = nullptr;            // optional device flag: set to 1 when an activation exceeds the fp16 planes' range (x3_range_check)
};
void launch_resblock1x3_fused(const RbFusedParams& p, hipStream_t s);
int rb_fused_tile(int CP);         // interior samples per workgroup (CP = 16 | 32 padded channels)
size_t rb_fused_w3_bytes(int KW, int CP);
// wp: packed fp32 weights [KW][CinP][CoutP] -> [K-step = (tap, 16-channel half)][plane][lane][8 fp16] for v_mfma_f32_32x32x16_f16
void launch_rb_pack_weights(const float* wp, int KW, int CinP, int CoutP, int CP, void* out, hipStream_t s);
void launch_add3_scale(const float* x0, const float* x1, const float* x2, float alpha, float* y, long long n, hipStream_t s);

}  // namespace dtts
