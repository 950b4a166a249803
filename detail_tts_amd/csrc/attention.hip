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

// MODE bits: 1 = T5 bias table, 2 = causal, 4 = VITS rel-key band
template <int D, int MODE>
__global__ __launch_bounds__(256) void flash_attn_kernel(const AttnParams p) {
    constexpr int VPITCH = D + 4;
    constexpr int DS = D / 4;     // k-steps of the QK^T product
    constexpr int CT = D / 16;    // 16-row tiles of the output channels
    constexpr bool DB = D <= 96;          // double-buffered LDS + register prefetch (does not fit for D = 192)
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int NLD = DB ? (D * KT) / 256 : 1;   // K (and V) elements staged per thread per tile
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ float smem[];
    float* Ks = smem;                              // [NBUF][D][KPITCH]
    float* Vt = Ks + NBUF * D * KPITCH;            // [NBUF][KT][VPITCH]
    float* bias_s = Vt + NBUF * KT * VPITCH;       // [129] (pre-multiplied by log2(e))

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware: the query blocks of one (sample, head) share K/V -> adjacent logical ids -> same XCD / L2
    const int nqb = (p.T + QPB - 1) / QPB;
    const int Lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = Lid % nqb, hb = Lid / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = qb * QPB;
    if (q0 >= len) return;

    const float* base = p.qkv + (long long)b * p.bs;
    const float* qp = base + (long long)(p.q_off + h * p.head_stride) * p.cs;
    const float* kp = base + (long long)(p.k_off + h * p.head_stride) * p.cs;
    const float* vp = base + (long long)(p.v_off + h * p.head_stride) * p.cs;

    if ((MODE & 1) && tid < 2 * BIAS_CLIP + 1) bias_s[tid] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + tid] * LOG2E;

    // Q fragments (B operand of S^T = K^T Q), pre-scaled by scale*log2(e) so the softmax runs on exp2
    const int tq0 = q0 + wave * QPW;
    const float qs = p.scale * LOG2E;
    float qreg[2][DS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int t = tq0 + qt * 16 + j;
        const int tc = t < len ? t : len - 1;
#pragma unroll
        for (int st = 0; st < DS; ++st) qreg[qt][st] = qp[(long long)(4 * st + g) * p.cs + tc] * qs;
    }

    floatx4 oacc[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) oacc[ct][qt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    const int klimit = (MODE & 2) ? min(len, q0 + QPB) : len;     // causal: keys beyond the block's last query never matter
    const int ntiles = (klimit + KT - 1) / KT;
    const bool wave_active = tq0 < len;

    // staging: thread owns elements idx = tid + 256*i of the [D][KT] tile -> (c = idx / KT, s = idx % KT); loads are
    // unconditional on clamped addresses (masking by select), issued one tile ahead and written to the other LDS buffer.
    float kreg[NLD], vreg[NLD];
    const int sl = tid & (KT - 1), c0 = tid / KT;                 // KT == 64, 256/64 = 4 rows per pass
    auto load_tile = [&](int kt) {
        const int s = kt * KT + sl;
        const int sc = s < len ? s : len - 1;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const long long off = (long long)(c0 + 4 * i) * p.cs + sc;
            kreg[i] = kp[off];
            vreg[i] = vp[off];
        }
    };
    auto store_tile = [&](int kt, int buf) {
        const bool ok = (kt * KT + sl) < len;
        float* kd = Ks + buf * D * KPITCH;
        float* vd = Vt + buf * KT * VPITCH;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            kd[(c0 + 4 * i) * KPITCH + sl] = kreg[i];
            vd[sl * VPITCH + c0 + 4 * i] = ok ? vreg[i] : 0.f;    // V must be finite where P == 0
        }
    };
    auto stage_direct = [&](int kt) {      // single-buffer path: global -> LDS without a register ring
        for (int idx = tid; idx < D * KT; idx += 256) {
            const int c = idx / KT, s = idx - c * KT;
            const bool ok = (kt * KT + s) < len;
            const long long off = (long long)c * p.cs + (ok ? kt * KT + s : len - 1);
            Ks[c * KPITCH + s] = kp[off];
            Vt[s * VPITCH + c] = ok ? vp[off] : 0.f;
        }
    };
    if (DB) {
        load_tile(0);
        store_tile(0, 0);
        __syncthreads();
    }

    for (int kt = 0; kt < ntiles; ++kt) {
        const int s0 = kt * KT, buf = DB ? (kt & 1) : 0;
        const bool has_next = kt + 1 < ntiles;
        if (DB) {
            if (has_next) load_tile(kt + 1);
        } else {
            __syncthreads();
            stage_direct(kt);
            __syncthreads();
        }
        if (wave_active) {
            const float* Kb = Ks + buf * D * KPITCH;
            const float* Vb = Vt + buf * KT * VPITCH;
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
                    const float a = Kb[(4 * st + g) * KPITCH + ks * 16 + j];
                    sacc[0][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qreg[0][st], sacc[0][ks], 0, 0, 0);
                    sacc[1][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qreg[1][st], sacc[1][ks], 0, 0, 0);
                }
            }
            // ---- bias, masks, online softmax in the log2 domain (per query column j of each q-tile)
            const bool full_tile = (s0 + KT <= len);
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
                        if (MODE & 1) {
                            int off = s - t;
                            off = off < -BIAS_CLIP ? -BIAS_CLIP : (off > BIAS_CLIP ? BIAS_CLIP : off);
                            v += bias_s[off + BIAS_CLIP];
                        }
                        if (MODE & 4) {
                            const int off = s - t;
                            if (off >= -p.band_w && off <= p.band_w && t < len)
                                v += LOG2E * p.band[(((long long)b * p.H + h) * p.T + t) * (2 * p.band_w + 1) + off + p.band_w];
                        }
                        if (MODE & 2) v = (s > t) ? -INFINITY : v;
                        if (!full_tile) v = (s >= len) ? -INFINITY : v;
                        sacc[qt][ks][r] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run[qt], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);     // m_run = -inf -> 0
                float sum = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(sacc[qt][ks][r] - m_use);
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
                        const float a = Vb[srow * VPITCH + ct * 16 + j];
                        oacc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sacc[0][ks][r], oacc[ct][0], 0, 0, 0);
                        oacc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sacc[1][ks][r], oacc[ct][1], 0, 0, 0);
                    }
                }
        }
        if (DB) {
            if (has_next) store_tile(kt + 1, buf ^ 1);
            __syncthreads();
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
            ml[0] = m_run[qt] * 0.6931471805599453f;       // back to the natural-log domain
            ml[1] = l_run[qt];
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(long long)(ct * 16 + 4 * g + r) * p.o_cs + t] = oacc[ct][qt][r] * inv;
    }
}

template <int D>
static void launch_d(const AttnParams& p, dim3 grid, size_t lds, hipStream_t stream) {
    const int mode = (p.bias_tab ? 1 : 0) | (p.causal ? 2 : 0) | (p.band ? 4 : 0);
    auto go = [&](auto kern, int) {
        if (lds > 64 * 1024) lds_optin(reinterpret_cast<const void*>(kern), (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    };
    switch (mode) {
        case 0: go(flash_attn_kernel<D, 0>, 0); break;
        case 1: go(flash_attn_kernel<D, 1>, 1); break;
        case 2: go(flash_attn_kernel<D, 2>, 2); break;
        case 4: go(flash_attn_kernel<D, 4>, 4); break;
        default: DTTS_REQUIRE(false, "unsupported attention mode combination");
    }
}

void launch_flash_attention(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.B > 0 && p.H > 0 && p.T > 0, "empty attention");
    dim3 grid(cdiv(p.T, QPB) * p.H * p.B);
    const char* tag = p.D == 48 ? "flash_attn_kernel<48>" : p.D == 64 ? "flash_attn_kernel<64>" : p.D == 96 ? "flash_attn_kernel<96>" : "flash_attn_kernel<192>";
    const double pairs = (double)p.B * p.H * (double)p.T * p.T * (p.causal ? 0.5 : 1.0);
    if (p.x3 && p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out) {
        static const bool old_kernel = []() { const char* v = getenv("DTTS_ATTN_KERNEL"); return v && v[0] == 'w'; }();
        ProfScope ps3(p.planes ? (old_kernel ? "flash_attn_x3w_kernel" : "flash_attn_x3b_kernel") : "flash_attn_x3_kernel<48>", 4.0 * pairs * p.D,
                      4.0 * (double)p.B * p.H * p.D * p.T * 4.0, stream);
        if (p.planes && old_kernel) launch_flash_attention_x3w(p, stream);
        else if (p.planes) launch_flash_attention_x3b(p, stream);
        else launch_flash_attention_x3(p, stream);
        return;
    }
    DTTS_REQUIRE(!p.out_x3, "split-precision attention output needs the x3 kernel (head dim 48, T5 bias)");
    ProfScope ps(tag, 4.0 * pairs * p.D, 4.0 * (double)p.B * p.H * p.D * p.T * 4.0, stream);
    auto lds = [](int D) { const int nb = D <= 96 ? 2 : 1; return sizeof(float) * (size_t)(nb * D * KPITCH + nb * KT * (D + 4) + 2 * BIAS_CLIP + 1); };
    switch (p.D) {
        case 48: launch_d<48>(p, grid, lds(48), stream); break;
        case 64: launch_d<64>(p, grid, lds(64), stream); break;
        case 96: launch_d<96>(p, grid, lds(96), stream); break;
        case 192: launch_d<192>(p, grid, lds(192), stream); break;
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
