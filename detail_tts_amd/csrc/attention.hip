// Flash-style fp32 attention over channel-major activations for gfx950.
//
//   out[b, h*D + c, t] = sum_s softmax_s( scale * sum_c q[c,t] k[c,s] + bias[h][clamp(s-t)] ) * v[c, s]
//
// q/k/v rows live in one [B, rows, T] tensor (the 1x1-conv output), addressed by
// (q_off|k_off|v_off) + h*head_stride + c, which covers both the reference's
// QKVAttentionLegacy layout (vqvae/utils/diff_util.py:155: per head [q|k|v]) and HF GPT-2's
// c_attn layout ([q|k|v] blocks of all heads).
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32):  the score tile is computed TRANSPOSED,
// S^T[s,t] = K^T Q, so that in the C layout (col = lane&15 = query, rows = 4*(lane>>4)+reg = key)
// a query's scores sit in 16 registers of 4 lanes -> the online-softmax row reductions are
// in-register plus two xor-shuffles, and P^T is *already* the B operand of the PV product
// O[c,t] += V[c,s] P^T[s,t] (k-slot g of step (ks,reg) <-> key ks*16+4g+reg), so P never
// leaves registers.  K is staged K-major [D][64] (pitch 80) and V transposed [64][D+4] in LDS
// so both A-operand reads are conflict-free ds_read_b32.
#include "attention.h"
#include "prof.h"

namespace dtts {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int KT = 64;        // keys per LDS tile
constexpr int KPITCH = 80;    // K tile row pitch (== 16 mod 32)
constexpr int QPW = 32;       // queries per wave (2 MFMA column tiles)
constexpr int QPB = 128;      // queries per block
constexpr int BIAS_CLIP = 64; // |s-t| beyond this shares one bucket (RelativePositionBias max_distance)

template <int D>
__global__ __launch_bounds__(256) void flash_attn_kernel(const AttnParams p) {
    constexpr int VPITCH = D + 4;
    constexpr int DS = D / 4;     // k-steps of the QK^T product
    constexpr int CT = D / 16;    // 16-row tiles of the output channels
    extern __shared__ float smem[];
    float* Ks = smem;                       // [D][KPITCH]
    float* Vt = Ks + D * KPITCH;            // [KT][VPITCH]
    float* bias_s = Vt + KT * VPITCH;       // [129]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = blockIdx.x * QPB;
    if (q0 >= len) return;

    const float* base = p.qkv + (long long)b * p.bs;
    const float* qp = base + (long long)(p.q_off + h * p.head_stride) * p.cs;
    const float* kp = base + (long long)(p.k_off + h * p.head_stride) * p.cs;
    const float* vp = base + (long long)(p.v_off + h * p.head_stride) * p.cs;

    if (p.bias_tab && tid < 2 * BIAS_CLIP + 1) bias_s[tid] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + tid];

    // Q fragments: B operand of S^T = K^T Q : lane (kq=g, j) holds Q[c = 4*step+g][t]
    const int tq0 = q0 + wave * QPW;
    float qreg[2][DS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int t = tq0 + qt * 16 + j;
#pragma unroll
        for (int st = 0; st < DS; ++st) qreg[qt][st] = (t < len) ? qp[(long long)(4 * st + g) * p.cs + t] * p.scale : 0.f;
    }

    floatx4 oacc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) oacc[ct][qt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    // causal: keys beyond the block's last query are never needed
    const int klimit = p.causal ? min(len, q0 + QPB) : len;
    const int ntiles = (klimit + KT - 1) / KT;
    const bool wave_active = tq0 < len;

    for (int kt = 0; kt < ntiles; ++kt) {
        const int s0 = kt * KT;
        __syncthreads();   // previous tile fully consumed
        // stage K [D][KT] and V^T [KT][D]
        for (int idx = tid; idx < D * KT; idx += 256) {
            const int c = idx / KT, s = idx - c * KT;
            const bool ok = (s0 + s) < len;
            const float kvl = ok ? kp[(long long)c * p.cs + s0 + s] : 0.f;
            const float vvl = ok ? vp[(long long)c * p.cs + s0 + s] : 0.f;
            Ks[c * KPITCH + s] = kvl;
            Vt[s * VPITCH + c] = vvl;
        }
        __syncthreads();
        if (!wave_active) continue;

        // ---- S^T = K^T Q
        floatx4 sacc[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sacc[qt][ks] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < DS; ++st) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float a = Ks[(4 * st + g) * KPITCH + ks * 16 + j];
                sacc[0][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qreg[0][st], sacc[0][ks], 0, 0, 0);
                sacc[1][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qreg[1][st], sacc[1][ks], 0, 0, 0);
            }
        }

        // ---- bias, masks, online softmax (per query column j of each q-tile)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int t = tq0 + qt * 16 + j;
            float mx = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int s = s0 + ks * 16 + 4 * g + r;
                    float v = sacc[qt][ks][r];
                    if (p.bias_tab) {
                        int off = s - t;
                        off = off < -BIAS_CLIP ? -BIAS_CLIP : (off > BIAS_CLIP ? BIAS_CLIP : off);
                        v += bias_s[off + BIAS_CLIP];
                    }
                    if (p.band) {
                        const int off = s - t;
                        if (off >= -p.band_w && off <= p.band_w && t < len)
                            v += p.band[(((long long)b * p.H + h) * p.T + t) * (2 * p.band_w + 1) + off + p.band_w];
                    }
                    const bool dead = (s >= len) || (p.causal && s > t);
                    v = dead ? -INFINITY : v;
                    sacc[qt][ks][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = expf(m_run[qt] - m_use);     // m_run=-inf -> 0
            float sum = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = expf(sacc[qt][ks][r] - m_use);
                    sacc[qt][ks][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            l_run[qt] = l_run[qt] * alpha + sum;
            m_run[qt] = m_new;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) oacc[ct][qt] *= alpha;
        }

        // ---- O += V P^T : A = V[c][key], B = P^T (registers)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int srow = ks * 16 + 4 * g + r;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float a = Vt[srow * VPITCH + ct * 16 + j];
                    oacc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sacc[0][ks][r], oacc[ct][0], 0, 0, 0);
                    oacc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sacc[1][ks][r], oacc[ct][1], 0, 0, 0);
                }
            }
    }

    if (!wave_active) return;
    float* ob = p.out + (long long)b * p.o_bs + (long long)(h * D) * p.o_cs;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int t = tq0 + qt * 16 + j;
        if (t >= len) continue;
        const float inv = 1.f / l_run[qt];
        if (p.ml_out && g == 0) {
            float* ml = p.ml_out + (((long long)b * p.H + h) * p.T + t) * 2;
            ml[0] = m_run[qt];
            ml[1] = l_run[qt];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long long)(ct * 16 + 4 * g + r) * p.o_cs + t] = oacc[ct][qt][r] * inv;
    }
}

void launch_flash_attention(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.B > 0 && p.H > 0 && p.T > 0, "empty attention");
    dim3 grid(cdiv(p.T, QPB), p.H, p.B);
    const char* tag = p.D == 48 ? "flash_attn_kernel<48>" : p.D == 64 ? "flash_attn_kernel<64>" : p.D == 96 ? "flash_attn_kernel<96>" : "flash_attn_kernel<192>";
    const double pairs = (double)p.B * p.H * (double)p.T * p.T * (p.causal ? 0.5 : 1.0);
    ProfScope ps(tag, 4.0 * pairs * p.D, 4.0 * (double)p.B * p.H * p.D * p.T * 4.0, stream);
    auto lds = [](int D) { return sizeof(float) * (size_t)(D * KPITCH + KT * (D + 4) + 2 * BIAS_CLIP + 1); };
    switch (p.D) {
        case 48: hipLaunchKernelGGL(flash_attn_kernel<48>, grid, dim3(256), lds(48), stream, p); break;
        case 64: hipLaunchKernelGGL(flash_attn_kernel<64>, grid, dim3(256), lds(64), stream, p); break;
        case 96: hipLaunchKernelGGL(flash_attn_kernel<96>, grid, dim3(256), lds(96), stream, p); break;
        case 192: {
            static bool once = false;
            if (!once) {
                DTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<192>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds(192)));
                once = true;
            }
            hipLaunchKernelGGL(flash_attn_kernel<192>, grid, dim3(256), lds(192), stream, p);
            break;
        }
        default: DTTS_REQUIRE(false, "unsupported head dim (48, 64, 96, 192)");
    }
    DTTS_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------
// VITS relative-position terms (banded, |s-t| <= W)
// ------------------------------------------------------------------------------------------
__global__ void vits_rel_key_kernel(const float* qkv, long long bs, int cs, int q_off, int head_stride, const float* Ek,
                                    float* relk, const int* lens, int H, int D, int T, int W, float scale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    if (t >= len) return;
    const float* qp = qkv + (long long)b * bs + (long long)(q_off + h * head_stride) * cs + t;
    const int R = 2 * W + 1;
    float acc[16];
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c = 0; c < D; ++c) {
        const float q = qp[(long long)c * cs];
        for (int r = 0; r < R; ++r) acc[r] += q * Ek[r * D + c];
    }
    float* o = relk + (((long long)b * H + h) * T + t) * R;
    for (int r = 0; r < R; ++r) o[r] = acc[r] * scale;
}

__global__ void vits_rel_value_kernel(const float* qkv, long long bs, int cs, int q_off, int k_off, int head_stride,
                                      const float* relk, const float* ml, const float* Ev, float* out, long long o_bs,
                                      int o_cs, const int* lens, int H, int D, int T, int W, float scale) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    if (t >= len) return;
    const float* base = qkv + (long long)b * bs;
    const float* qp = base + (long long)(q_off + h * head_stride) * cs;
    const float* kp = base + (long long)(k_off + h * head_stride) * cs;
    const int R = 2 * W + 1;
    const long long qi = ((long long)b * H + h) * T + t;
    const float m = ml[qi * 2], l = ml[qi * 2 + 1];
    float pr[16];
    for (int r = 0; r < R; ++r) {
        const int s = t + r - W;
        float p = 0.f;
        if (s >= 0 && s < len) {
            float dot = 0.f;
            for (int c = 0; c < D; ++c) dot += qp[(long long)c * cs + t] * kp[(long long)c * cs + s];
            p = expf(dot * scale + relk[qi * R + r] - m) / l;
        }
        pr[r] = p;
    }
    float* ob = out + (long long)b * o_bs + (long long)(h * D) * o_cs + t;
    for (int c = 0; c < D; ++c) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += pr[r] * Ev[r * D + c];
        ob[(long long)c * o_cs] += a;
    }
}

void launch_vits_rel_key(const float* qkv, long long bs, int cs, int q_off, int head_stride, const float* Ek,
                         float* relk, const int* lens, int B, int H, int D, int T, int W, float scale, hipStream_t s) {
    DTTS_REQUIRE(2 * W + 1 <= 16, "window");
    dim3 grid(cdiv(T, 128), H, B);
    hipLaunchKernelGGL(vits_rel_key_kernel, grid, dim3(128), 0, s, qkv, bs, cs, q_off, head_stride, Ek, relk, lens, H, D, T, W, scale);
    DTTS_CHECK_HIP(hipGetLastError());
}

void launch_vits_rel_value(const float* qkv, long long bs, int cs, int q_off, int k_off, int head_stride,
                           const float* relk, const float* ml, const float* Ev, float* out, long long o_bs, int o_cs,
                           const int* lens, int B, int H, int D, int T, int W, float scale, hipStream_t s) {
    DTTS_REQUIRE(2 * W + 1 <= 16, "window");
    dim3 grid(cdiv(T, 128), H, B);
    hipLaunchKernelGGL(vits_rel_value_kernel, grid, dim3(128), 0, s, qkv, bs, cs, q_off, k_off, head_stride, relk, ml, Ev,
                       out, o_bs, o_cs, lens, H, D, T, W, scale);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
