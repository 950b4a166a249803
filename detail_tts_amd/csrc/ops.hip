// Small HBM-bound kernels (see ops.h).  All are launched with >= 256-thread blocks and coalesce
// along the contiguous time axis of the [B, C, T] layout; reductions use wave64 shuffles.
#include <atomic>
#include "ops.h"
#include "philox.h"

namespace dtts {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // red: >= 4 floats (256 threads)
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// ---------------------------------------------------------------------------------------------
// One pass over the group: shifted sums S1 = sum(x - k), S2 = sum((x - k)^2) with k = first element (no cancellation
// problem even when |mean| >> std), float4 loads when the rows are 16-byte aligned.
__global__ __launch_bounds__(256) void gn_coeffs_kernel(const float* x, long long x_bs, int x_cs, const int* lens, int T, int C,
                                                        int groups, const float* gamma, const float* beta, float eps,
                                                        const float* ada, int ada_stride, int ada_bs, float* ab_out, const int* ada_idx) {
    __shared__ float red[8];
    const int g = blockIdx.x, b = blockIdx.y;
    const int len = lens ? lens[b] : T;
    const int cpg = C / groups;
    const float* xg = x + (long long)b * x_bs + (long long)(g * cpg) * x_cs;
    const int n = cpg * len;
    const float k = xg[0];
    float s1 = 0.f, s2 = 0.f;
    const bool vec = ((x_cs & 3) == 0) && ((reinterpret_cast<unsigned long long>(xg) & 15ull) == 0);
    if (vec) {
        const int len4 = len >> 2;
        // 8 rows per pass: 8 independent 16-byte loads in flight per thread
        int c = 0;
        for (; c + 8 <= cpg; c += 8) {
            for (int t = threadIdx.x; t < len4; t += blockDim.x) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(xg + (long long)(c + u) * x_cs)[t];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float d0 = v[u].x - k, d1 = v[u].y - k, d2 = v[u].z - k, d3 = v[u].w - k;
                    s1 += (d0 + d1) + (d2 + d3);
                    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            }
        }
        for (; c < cpg; ++c) {
            const float4* row = reinterpret_cast<const float4*>(xg + (long long)c * x_cs);
            for (int t = threadIdx.x; t < len4; t += blockDim.x) {
                const float4 v = row[t];
                const float d0 = v.x - k, d1 = v.y - k, d2 = v.z - k, d3 = v.w - k;
                s1 += (d0 + d1) + (d2 + d3);
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        for (int cc = 0; cc < cpg; ++cc)
            for (int t = (len4 << 2) + threadIdx.x; t < len; t += blockDim.x) {
                const float d = xg[(long long)cc * x_cs + t] - k;
                s1 += d;
                s2 += d * d;
            }
    } else {
        for (int c = 0; c < cpg; ++c)
            for (int t = threadIdx.x; t < len; t += blockDim.x) {
                const float d = xg[(long long)c * x_cs + t] - k;
                s1 += d;
                s2 += d * d;
            }
    }
    const float S1 = block_sum(s1, red) / (float)n;
    const float S2 = block_sum(s2, red) / (float)n;
    const float mean = k + S1;
    const float var = fmaxf(S2 - S1 * S1, 0.f);
    const float rstd = rsqrtf(var + eps);
    if (threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        float a = rstd * gamma[c];
        float d = beta[c] - mean * a;
        if (ada) {
            const float* ad = ada + (ada_idx ? (long long)ada_idx[b] : (long long)b * ada_bs);
            const float sc = 1.f + ad[(long long)c * ada_stride], sh = ad[(long long)(C + c) * ada_stride];
            a *= sc;
            d = d * sc + sh;
        }
        float* o = ab_out + ((long long)b * C + c) * 2;
        o[0] = a;
        o[1] = d;
    }
}

void launch_gn_coeffs(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int groups,
                      const float* gamma, const float* beta, float eps, const float* ada, int ada_stride, int ada_bs,
                      float* ab_out, hipStream_t s, const int* ada_idx) {
    DTTS_REQUIRE(C % groups == 0 && C / groups <= 256, "group size");
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3(groups, B), dim3(256), 0, s, x, x_bs, x_cs, lens, T, C, groups, gamma, beta, eps,
                       ada, ada_stride, ada_bs, ab_out, ada_idx);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
__global__ void affine_apply_kernel(const float* x, long long x_bs, int x_cs, const float* ab, const int* lens, int T, int C,
                                    int up, int act, float* y, long long y_bs, int y_cs) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int len = (lens ? lens[b] : T) * up;
    const float a = ab ? ab[((long long)b * C + c) * 2] : 1.f, d = ab ? ab[((long long)b * C + c) * 2 + 1] : 0.f;
    const float* xr = x + (long long)b * x_bs + (long long)c * x_cs;
    float* yr = y + (long long)b * y_bs + (long long)c * y_cs;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < len; t += gridDim.x * blockDim.x)
        yr[t] = act_apply(a * xr[t / up] + d, act, 0.f);
}

void launch_affine_apply(const float* x, long long x_bs, int x_cs, const float* ab, const int* lens, int T, int B, int C,
                         int up, int act, float* y, long long y_bs, int y_cs, hipStream_t s) {
    dim3 grid(cdiv(T * up, 256) > 8 ? 8 : cdiv(T * up, 256), C, B);
    hipLaunchKernelGGL(affine_apply_kernel, grid, dim3(256), 0, s, x, x_bs, x_cs, ab, lens, T, C, up, act, y, y_bs, y_cs);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels (two-pass: mean, then centred sum of squares).  A 256-thread workgroup owns 16 time columns:
// thread (tx = column, ty = one of 16 channel groups) strides over C, the 16 partial sums per column are combined through LDS
// in a fixed order.  Loads stay coalesced along the contiguous time axis (16 x 4 B segments).
__global__ __launch_bounds__(256) void ln_channels_kernel(const float* x, const float* r, long long bs, int cs, const int* lens, int T, int C,
                                                          const float* gamma, const float* beta, float eps, float* y, long long y_bs, int y_cs) {
    __shared__ float red[16][17];
    const int b = blockIdx.y, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tx;
    const int len = lens ? lens[b] : T;
    const bool ok = t < len;
    const int tc = ok ? t : (len > 0 ? len - 1 : 0);
    const float* xb = x + (long long)b * bs + tc;
    const float* rb = r ? r + (long long)b * bs + tc : nullptr;
    float s = 0.f;
    for (int c = ty; c < C; c += 16) s += xb[(long long)c * cs] + (rb ? rb[(long long)c * cs] : 0.f);
    red[ty][tx] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i][tx];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = ty; c < C; c += 16) {
        const float d = xb[(long long)c * cs] + (rb ? rb[(long long)c * cs] : 0.f) - mean;
        q += d * d;
    }
    red[ty][tx] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) qt += red[i][tx];
    const float rstd = rsqrtf(qt / (float)C + eps);
    if (!ok) return;
    float* yb = y + (long long)b * y_bs + t;
    for (int c = ty; c < C; c += 16) {
        const float v = xb[(long long)c * cs] + (rb ? rb[(long long)c * cs] : 0.f);
        yb[(long long)c * y_cs] = (v - mean) * rstd * gamma[c] + beta[c];
    }
}

// The same LayerNorm with a thread's channels held in registers (C <= 16 N): ONE pass of loads, all in flight at once, instead of three
// passes of dependent strided loads (the GPT prefill's 1024-channel norms: 33 -> ~10 us per launch).  Every sum runs over the same
// values in the same order as ln_channels_kernel, so the two are interchangeable bit for bit.
template <int N>
__global__ __launch_bounds__(256) void ln_channels_reg_kernel(const float* x, const float* r, long long bs, int cs, const int* lens, int T, int C,
                                                              const float* gamma, const float* beta, float eps, float* y, long long y_bs,
                                                              int y_cs) {
    __shared__ float red[16][17];
    const int b = blockIdx.y, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + tx;
    const int len = lens ? lens[b] : T;
    const bool ok = t < len;
    const int tc = ok ? t : (len > 0 ? len - 1 : 0);
    const float* xb = x + (long long)b * bs + tc;
    const float* rb = r ? r + (long long)b * bs + tc : nullptr;
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = ty + 16 * i;
        v[i] = c < C ? xb[(long long)c * cs] + (rb ? rb[(long long)c * cs] : 0.f) : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (ty + 16 * i < C) s += v[i];
    red[ty][tx] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i][tx];
    const float mean = tot / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (ty + 16 * i < C) {
            const float d = v[i] - mean;
            q += d * d;
        }
    red[ty][tx] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) qt += red[i][tx];
    const float rstd = rsqrtf(qt / (float)C + eps);
    if (!ok) return;
    float* yb = y + (long long)b * y_bs + t;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = ty + 16 * i;
        if (c < C) yb[(long long)c * y_cs] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

static std::atomic<int> g_ln_reg{[]() { const char* v = getenv("DTTS_LN_REG"); return (v && v[0] == '0') ? 0 : 1; }()};
void set_ln_channels_reg(bool on) { g_ln_reg.store(on ? 1 : 0, std::memory_order_relaxed); }   // option "ln_reg" (process-wide)

void launch_ln_channels(const float* x, const float* r, long long bs, int cs, const int* lens, int T, int B, int C,
                        const float* gamma, const float* beta, float eps, float* y, long long y_bs, int y_cs, hipStream_t s) {
    const bool reg_on = g_ln_reg.load(std::memory_order_relaxed) != 0;
    if (reg_on && C <= 16 * 16) {
        hipLaunchKernelGGL(ln_channels_reg_kernel<16>, dim3(cdiv(T, 16), B), dim3(256), 0, s, x, r, bs, cs, lens, T, C, gamma, beta, eps, y, y_bs, y_cs);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (reg_on && C <= 16 * 64) {
        hipLaunchKernelGGL(ln_channels_reg_kernel<64>, dim3(cdiv(T, 16), B), dim3(256), 0, s, x, r, bs, cs, lens, T, C, gamma, beta, eps, y, y_bs, y_cs);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(ln_channels_kernel, dim3(cdiv(T, 16), B), dim3(256), 0, s, x, r, bs, cs, lens, T, C, gamma, beta, eps, y,
                       y_bs, y_cs);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_time_kernel(const float* x, long long x_bs, int x_cs, const int* lens, int T, int C,
                                                        float* out) {
    __shared__ float red[8];
    const int c = blockIdx.x, b = blockIdx.y;
    const int len = lens ? lens[b] : T;
    const float* xr = x + (long long)b * x_bs + (long long)c * x_cs;
    float s = 0.f;
    for (int t = threadIdx.x; t < len; t += blockDim.x) s += xr[t];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[(long long)b * C + c] = s / (float)len;
}

void launch_mean_time(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_time_kernel, dim3(C, B), dim3(256), 0, s, x, x_bs, x_cs, lens, T, C, out);
    DTTS_CHECK_HIP(hipGetLastError());
}

__global__ void broadcast_channels_kernel(const float* v, int C, int T, float* y, long long y_bs, int y_cs) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float val = v[c];
    float* yr = y + (long long)b * y_bs + (long long)c * y_cs;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) yr[t] = val;
}

void launch_broadcast_channels(const float* v, int B, int C, int T, float* y, long long y_bs, int y_cs, hipStream_t s) {
    hipLaunchKernelGGL(broadcast_channels_kernel, dim3(cdiv(T, 256) > 8 ? 8 : cdiv(T, 256), C, B), dim3(256), 0, s, v, C, T, y,
                       y_bs, y_cs);
    DTTS_CHECK_HIP(hipGetLastError());
}

__global__ void timestep_sinusoid_kernel(const int* ts, int n, int dim, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over half*n
    const int half = dim / 2;
    if (i >= half * n) return;
    const int c = i / n, k = i - c * n;
    const float freq = expf(-logf(10000.f) * (float)c / (float)half);
    const float arg = (float)ts[k] * freq;
    out[(long long)c * n + k] = cosf(arg);
    out[(long long)(half + c) * n + k] = sinf(arg);
}

void launch_timestep_sinusoid(const int* ts, int n, int dim, float* out, hipStream_t s) {
    hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3(cdiv(dim / 2 * n, 256)), dim3(256), 0, s, ts, n, dim, out);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
__global__ void philox_normal_kernel(float* out, long long bs, int n, unsigned long long seed, const int* sample_ids, int stage,
                                     int step, float scale) {
    const int b = blockIdx.y;
    const unsigned sample = (unsigned)sample_ids[b];
    float* ob = out + (long long)b * bs;
    const int nblk = (n + 3) / 4;
    for (int blk = blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += gridDim.x * blockDim.x) {
        float z[4];
        philox_normal4(seed, sample, stage, step, (unsigned)blk, z);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (blk * 4 + i < n) ob[blk * 4 + i] = z[i] * scale;
    }
}

void launch_philox_normal(float* out, long long bs, int n, int B, unsigned long long seed, const int* sample_ids, int stage,
                          int step, float scale, hipStream_t s) {
    const int nblk = (n + 3) / 4;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(cdiv(nblk, 256) > 64 ? 64 : cdiv(nblk, 256), B), dim3(256), 0, s, out, bs, n,
                       seed, sample_ids, stage, step, scale);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Sampler update.  Noise element index inside a sample follows the reference tensor order [C, T_b]
// with T_b the sample's own length (a sample's noise never depends on batch padding).
__global__ void diff_update_kernel(float* x, long long x_bs, int x_cs, const float* mo, long long m_bs, int m_cs, const int* lens,
                                   int T, int B, int C, DiffStepCoefs k, unsigned long long seed, const int* sample_ids, int step,
                                   const float* noise_override, int final_denorm, float* x0_out) {
    const int b = blockIdx.y;
    const int len = lens ? lens[b] : T;
    const unsigned sample = (unsigned)sample_ids[b];
    float* xb = x + (long long)b * x_bs;
    const float* mc = mo + (long long)b * m_bs;
    const float* mu = mo + (long long)(B + b) * m_bs;
    const int n = C * len;
    const int nblk = (n + 3) / 4;
    for (int blk = blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += gridDim.x * blockDim.x) {
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (k.nonzero && !noise_override) philox_normal4(seed, sample, STAGE_DIFF_STEP, step, (unsigned)blk, z);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = blk * 4 + i;
            if (e >= n) break;
            const int c = e / len, t = e - c * len;
            const float xv = xb[(long long)c * x_cs + t];
            const float eps_c = mc[(long long)c * m_cs + t], vv = mc[(long long)(C + c) * m_cs + t];
            const float eps_u = mu[(long long)c * m_cs + t];
            const float frac = (vv + 1.f) * 0.5f;
            const float log_var = frac * k.max_log + (1.f - frac) * k.min_log;
            const float eps = (1.f + k.cfk) * eps_c - k.cfk * eps_u;
            float x0 = k.sqrt_recip_ac * xv - k.sqrt_recipm1_ac * eps;
            x0 = fminf(fmaxf(x0, -1.f), 1.f);
            if (x0_out) x0_out[(long long)b * C * T + (long long)c * T + t] = x0;
            float v = k.coef1 * x0 + k.coef2 * xv;
            if (k.nonzero) {
                const float nz = noise_override ? noise_override[(long long)b * C * T + (long long)c * T + t] : z[i];
                v += expf(0.5f * log_var) * nz;
            }
            if (final_denorm) v = ((v + 1.f) * 0.5f) * (2.7f - (-11.512925465f)) + (-11.512925465f);
            xb[(long long)c * x_cs + t] = v;
        }
    }
}

void launch_diff_update(float* x, long long x_bs, int x_cs, const float* model_out, long long m_bs, int m_cs, const int* lens,
                        int T, int B, int C, DiffStepCoefs k, unsigned long long seed, const int* sample_ids, int step,
                        const float* noise_override, int final_denorm, hipStream_t s, float* x0_out) {
    const int nblk = (C * T + 3) / 4;
    hipLaunchKernelGGL(diff_update_kernel, dim3(cdiv(nblk, 256) > 64 ? 64 : cdiv(nblk, 256), B), dim3(256), 0, s, x, x_bs, x_cs,
                       model_out, m_bs, m_cs, lens, T, B, C, k, seed, sample_ids, step, noise_override, final_denorm, x0_out);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
__global__ void flow_prior_kernel(const float* stats, long long s_bs, int s_cs, const int* lens, int T, int C, float noise_scale,
                                  unsigned long long seed, const int* sample_ids, const float* noise_override, int flip,
                                  float* z, long long z_bs, int z_cs) {
    const int b = blockIdx.y;
    const int len = lens ? lens[b] : T;
    const unsigned sample = (unsigned)sample_ids[b];
    const float* sb = stats + (long long)b * s_bs;
    float* zb = z + (long long)b * z_bs;
    const int n = C * len, nblk = (n + 3) / 4;
    for (int blk = blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += gridDim.x * blockDim.x) {
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (!noise_override) philox_normal4(seed, sample, STAGE_FLOW_PRIOR, 0, (unsigned)blk, nz);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = blk * 4 + i;
            if (e >= n) break;
            const int c = e / len, t = e - c * len;
            const float m = sb[(long long)c * s_cs + t], logs = sb[(long long)(C + c) * s_cs + t];
            const float nv = noise_override ? noise_override[(long long)b * C * T + (long long)c * T + t] : nz[i];
            zb[(long long)(flip ? C - 1 - c : c) * z_cs + t] = m + nv * expf(logs) * noise_scale;
        }
    }
}

void launch_flow_prior(const float* stats, long long s_bs, int s_cs, const int* lens, int T, int B, int C, float noise_scale,
                       unsigned long long seed, const int* sample_ids, const float* noise_override, int flip, float* z,
                       long long z_bs, int z_cs, hipStream_t s) {
    const int nblk = (C * T + 3) / 4;
    hipLaunchKernelGGL(flow_prior_kernel, dim3(cdiv(nblk, 256) > 64 ? 64 : cdiv(nblk, 256), B), dim3(256), 0, s, stats, s_bs,
                       s_cs, lens, T, C, noise_scale, seed, sample_ids, noise_override, flip, z, z_bs, z_cs);
    DTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// y = flip_channels( cat(x0, x1 - m) )  with x = [x0 | x1], half = Ctot/2.  flip=0 skips the flip.
__global__ void coupling_reverse_kernel(const float* x, const float* m, float* y, long long bs, int cs, const int* lens, int T,
                                        int Ctot, int flip) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    const int half = Ctot / 2;
    const float* xr = x + (long long)b * bs + (long long)c * cs;
    const float* mr = (c >= half) ? m + (long long)b * (long long)half * cs + (long long)(c - half) * cs : nullptr;
    const int co = flip ? (Ctot - 1 - c) : c;
    float* yr = y + (long long)b * bs + (long long)co * cs;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < len; t += gridDim.x * blockDim.x)
        yr[t] = mr ? xr[t] - mr[t] : xr[t];
}

void launch_coupling_reverse(const float* x, const float* m, float* y, long long bs, int cs, const int* lens, int T, int B,
                             int Ctot, int flip, hipStream_t s) {
    hipLaunchKernelGGL(coupling_reverse_kernel, dim3(cdiv(T, 256) > 8 ? 8 : cdiv(T, 256), Ctot, B), dim3(256), 0, s, x, m, y, bs,
                       cs, lens, T, Ctot, flip);
    DTTS_CHECK_HIP(hipGetLastError());
}

__global__ void add3_scale_kernel(const float* x0, const float* x1, const float* x2, float alpha, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = alpha * (x0[i] + x1[i] + x2[i]);
}

void launch_add3_scale(const float* x0, const float* x1, const float* x2, float alpha, float* y, long long n, hipStream_t s) {
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(add3_scale_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, x0, x1, x2, alpha, y, n);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
