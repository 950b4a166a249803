// Split-precision (2 x fp16 planes, 3 products) flash attention for the diffusion trunk (head dim 48, T5 relative-position bias).
//
// Same algorithm and register mapping as flash_attn_kernel (attention.hip): the score tile is computed transposed,
// S^T[key, query] = K^T Q, so each query's scores sit in registers of 4 lanes and P^T is already the B operand of
// O[c, query] += V[c, key] P^T[key, query].  The matrix products run on v_mfma_f32_16x16x32_f16 with every fp32 operand (scaled by a
// power of two) split into two fp16 planes (a = h0 + h1, 22 bits) and the three cross products h0 h0', h0 h1', h1 h0' accumulated
// in fp32 (conv_x3.h) - fp32-GEMM-class error at 42 instead of 192 matrix instructions per 16 x 64 score tile.
//
//   scales: Q carries scale * log2(e) * 16, K carries 16 -> the score accumulator is 256 S (undone by the fma that adds the bias);
//           P = exp2(s - m + 10) (the 1024 is free in the exponent), V carries 16 -> the output accumulator is 16384 O, and the
//           denominator is accumulated from the same scaled P: O = acc / (16 l').
//   QK^T : K = 48 channels = one full 32-channel MFMA + one half-used (channels 48..63 of Q are zero, the K lanes of that half
//          re-read valid chunks).  K tile in LDS: chunks (8 channels x 16 B) [plane][c8 0..5][key 0..63]   -> linear ds_read_b128
//   P V  : one MFMA consumes 32 keys = two 16-key score tiles; k-slot 8g + r <-> key 4g + r (first tile), 8g + 4 + r <-> key
//          16 + 4g + r (second).  V tile in LDS: chunks (8 keys in that slot order) [plane][ct][u][g][channel 0..15]
// K, V and Q are split on the fly while staging (hardware v_cvt_pk_f16_f32); P is split in registers after the softmax.
#include "attention.h"
#include "conv_x3.h"
#include "split3.h"

namespace dtts {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D = 48, KT = 64, NW = 8, QPW = 16, QPB = NW * QPW, BIAS_CLIP = 64;
constexpr int KCH = 6 * KT;                    // K chunks per plane
constexpr int VCH = 3 * 2 * 4 * 16;            // V chunks per plane
constexpr int NPL = XS_PLANES;
constexpr int BUF_BYTES = NPL * (KCH + VCH) * 16;   // 24 KiB per stage
constexpr float QK_SCALE = 16.f, P_SHIFT = 10.f, V_SCALE = 16.f;
constexpr float M_SLACK = 3.f;                      // the running max is raised only when a tile exceeds it by more than 2^3 (P <= 8192)
constexpr int NBUF = 2;                            // LDS stages (1: single buffer, two barriers per tile, half the LDS)

__device__ __forceinline__ hf8 as_hf(const uint4& q) { return __builtin_bit_cast(hf8, q); }
__device__ __forceinline__ hf4 as_hf4(const uint2& q) { return __builtin_bit_cast(hf4, q); }

// the three significant cross products, smallest first
#define DTTS_X3_MFMA(acc, A, Bq)                                                         \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1], Bq[0], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], Bq[1], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0], Bq[0], acc, 0, 0, 0);

// 512 threads = 8 waves x 16 queries: <= 128 VGPRs -> 2 workgroups (16 waves) per CU share each staged K/V tile 8 ways
// This kernel takes fp32 q / k / v and splits them while staging (unit entry points, DTTS_ATTN_PLANES=0); the trunk's default path
// hands over AttnPlanes operand images and runs attention_x3w.hip.
__global__ __launch_bounds__(512, 4) void flash_attn_x3_kernel(const AttnParams p) {
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_s = reinterpret_cast<float*>(smem + NBUF * BUF_BYTES);      // [129], pre-multiplied by log2(e)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
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
    if (tid < 2 * BIAS_CLIP + 1) bias_s[tid] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + tid] * LOG2E;

    // ---- Q fragments: B operand, lane (query j, g) holds channels kb*32 + 8g .. +7 of each plane, pre-scaled by scale*log2(e)*16
    const int tq0 = q0 + wave * QPW;
    const float qs = p.scale * LOG2E * QK_SCALE;
    hf8 qf0[NPL], qf1[NPL];           // channels 8g .. 8g+7 and 32 + 8g .. +7 (zero beyond channel 47)
    {
        const int t = tq0 + j;
        const int tc = t < len ? t : len - 1;
        float v[8];
        uint4 w0, w1;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = qp[(long long)(8 * g + e) * p.cs + tc] * qs;
        split8(v, w0, w1);
        qf0[0] = as_hf(w0);
        qf0[1] = as_hf(w1);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g < 2 ? qp[(long long)(32 + 8 * g + e) * p.cs + tc] * qs : 0.f;
        split8(v, w0, w1);
        qf1[0] = as_hf(w0);
        qf1[1] = as_hf(w1);
    }

    floatx4 oacc[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) oacc[ct] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY;
    const int ntiles = (len + KT - 1) / KT;
    const bool wave_active = tq0 < len;

    // ---- staging: K chunk (c8, key) = 8 channels of one key; V chunk (channel, octet o = u*4 + g) = keys {4g..4g+3, 16+4g..+3} + 32u.
    // 384 + 384 chunks per tile over 512 threads: threads < 384 own K chunk tid, threads >= 128 own V chunk tid - 128.
    const bool hasK = tid < 384, hasV = tid >= 128;
    const int kkey = tid & 63, kc8 = hasK ? tid >> 6 : 0;
    // V chunk v = ((ct*2 + u)*4 + g)*16 + c16 is also its LDS slot: a wave writes 64 consecutive chunks (conflict-free), and the 8
    // lanes (u, g) of one channel still consume whole 128-byte lines of its row
    const int vidx = hasV ? tid - 128 : 0;
    const int vc = ((vidx >> 7) << 4) + (vidx & 15);
    const int vkey = ((vidx >> 6) & 1) * 32 + ((vidx >> 4) & 3) * 4;       // first key of the octet inside the tile
    const bool vec_ok = ((p.cs & 3) == 0) && ((reinterpret_cast<unsigned long long>(vp) & 15ull) == 0);
    float kr[8], vr[8];
    auto load_tile = [&](int kt) {
        const int s0 = kt * KT;
        if (hasK) {
            const int s = s0 + kkey, sc = s < len ? s : len - 1;
#pragma unroll
            for (int e = 0; e < 8; ++e) kr[e] = kp[(long long)(kc8 * 8 + e) * p.cs + sc];
        }
        if (hasV) {
            if (vec_ok && (s0 + KT <= len)) {
                const float* r = vp + (long long)vc * p.cs + s0 + vkey;
                const float4 x0 = *reinterpret_cast<const float4*>(r), x1 = *reinterpret_cast<const float4*>(r + 16);
                vr[0] = x0.x; vr[1] = x0.y; vr[2] = x0.z; vr[3] = x0.w; vr[4] = x1.x; vr[5] = x1.y; vr[6] = x1.z; vr[7] = x1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sv = s0 + vkey + (e & 3) + (e >> 2) * 16;
                    const bool ok = sv < len;
                    const float a = vp[(long long)vc * p.cs + (ok ? sv : len - 1)];
                    vr[e] = ok ? a : 0.f;                                 // V must be finite (zero) where P == 0
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
        uint4* kd = reinterpret_cast<uint4*>(smem + buf * BUF_BYTES);     // [2][6][64]
        uint4* vd = kd + NPL * KCH;                                       // [2][ct][u][g][16]
        uint4 w0, w1;
        if (hasK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) kr[e] *= QK_SCALE;
            split8(kr, w0, w1);
            kd[0 * KCH + kc8 * KT + kkey] = w0; kd[1 * KCH + kc8 * KT + kkey] = w1;
        }
        if (hasV) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vr[e] *= V_SCALE;
            split8(vr, w0, w1);
            vd[0 * VCH + vidx] = w0; vd[1 * VCH + vidx] = w1;
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();

    // A-operand chunk column of this lane for the two channel blocks: kb 0 -> c8 = g; kb 1 -> c8 = 4 + (g & 1) (lanes g >= 2 meet
    // zero Q channels 48..63, any finite chunk will do)
    const int kcol0 = g * KT + j, kcol1 = (4 + (g & 1)) * KT + j;
    float l_lane = 0.f;             // this lane's share of the denominator (its 16 keys of every tile); combined over g at the end

    for (int kt = 0; kt < ntiles; ++kt) {
        const int s0 = kt * KT, buf = NBUF == 2 ? (kt & 1) : 0;
        const bool has_next = kt + 1 < ntiles;
        if (has_next) load_tile(kt + 1);
        if (wave_active) {
            const uint4* Kb = reinterpret_cast<const uint4*>(smem + buf * BUF_BYTES);
            const uint4* Vb = Kb + NPL * KCH;
            // The tile is processed in two independent 32-key halves (one PV MFMA step each): the second half's QK^T MFMAs carry
            // no dependence on the first half's softmax / split VALU work, so the two pipes overlap inside one wave.
            const bool full_tile = (s0 + KT <= len);
            const int t = tq0 + j;
            // every (key, query) pair of this wave's tile beyond the bias window on one side -> one bucket, no table look-ups
            const bool far_hi = s0 - (tq0 + QPW - 1) >= BIAS_CLIP, far_lo = (s0 + KT - 1) - tq0 <= -BIAS_CLIP;
            const bool far = (far_hi || far_lo) && full_tile;
            const float bfar = bias_s[far_hi ? 2 * BIAS_CLIP : 0];
            constexpr float SU = 1.f / (QK_SCALE * QK_SCALE);            // the score accumulator holds 256 S
            floatx4 sacc[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sacc[ks] = floatx4{0.f, 0.f, 0.f, 0.f};
                hf8 a[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Kb[pl * KCH + kcol0 + ks * 16]);
                DTTS_X3_MFMA(sacc[ks], a, qf0)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Kb[pl * KCH + kcol1 + ks * 16]);
                DTTS_X3_MFMA(sacc[ks], a, qf1)
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // exponent arguments e = s + bias (log2 domain) of this lane's 8 keys, and their maximum
                float mx = -INFINITY;
                if (far) {
                    float r = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) r = fmaxf(r, sacc[2 * u + q][i]);
                    mx = fmaf(r, SU, bfar);                              // SU > 0: the maximum commutes with the affine map
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int s = s0 + (2 * u + q) * 16 + 4 * g + r;
                            int off = s - t;
                            off = off < -BIAS_CLIP ? -BIAS_CLIP : (off > BIAS_CLIP ? BIAS_CLIP : off);
                            float v = fmaf(sacc[2 * u + q][r], SU, bias_s[off + BIAS_CLIP]);
                            if (!full_tile) v = (s >= len) ? -INFINITY : v;
                            sacc[2 * u + q][r] = v;
                            mx = fmaxf(mx, v);
                        }
                }
                // Lazy running maximum: raise it (exchange over the query's 4 lanes, rescale O and l) only when some query of this wave
                // sees a tile maximum more than 2^M_SLACK above it - rare after the first tiles; otherwise P = exp2(e - m + 10) <= 8192
                // stays far inside fp16's range and nothing needs rescaling.
                if (__builtin_amdgcn_ballot_w64(mx > m_run + M_SLACK) != 0ull) {
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    // per query: a query's sequence of maxima depends on its own scores only, never on its wave neighbours' (which
                    // may be padding) - results stay invariant to batch composition and to whatever the padded rows hold
                    const float m_new = (mx > m_run + M_SLACK) ? mx : m_run;
                    const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                    l_lane *= alpha;
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) oacc[ct] *= alpha;
                    m_run = m_new;
                }
                const float m_sub = ((m_run == -INFINITY) ? 0.f : m_run) - P_SHIFT;
                float pv[8];
                if (far) {
                    const float c0 = bfar - m_sub;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[q * 4 + r] = __builtin_amdgcn_exp2f(fmaf(sacc[2 * u + q][r], SU, c0));   // 1024 P: the scale is free
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[q * 4 + r] = __builtin_amdgcn_exp2f(sacc[2 * u + q][r] - m_sub);
                }
                l_lane += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
                hf8 pf[NPL];
                {
                    uint4 w0, w1;
                    split8_inrange(pv, w0, w1);
                    pf[0] = as_hf(w0);
                    pf[1] = as_hf(w1);
                }
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    hf8 a[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Vb[pl * VCH + ((ct * 2 + u) * 4 + g) * 16 + j]);
                    DTTS_X3_MFMA(oacc[ct], a, pf)
                }
            }
        }
        if (NBUF == 1) __syncthreads();                 // everyone is done reading the tile before it is overwritten
        if (has_next) store_tile(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }

    if (!wave_active) return;
    float l_run = l_lane;
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    const int t = tq0 + j;
    if (t >= len) return;
    const float inv = 1.f / (l_run * V_SCALE);          // acc = (1024 P)(16 V), l_run = sum of 1024 P
    if (p.out_x3) {
        // lane (j, g) holds channels ct*16 + 4g + r of query t: half (g & 1) of the 8-channel chunk c8 = h*6 + ct*2 + (g >> 1)
        unsigned char* ob = static_cast<unsigned char*>(p.out_x3) + ((long long)b * (p.H * D / 8) * NPL) * p.x3_tp * 16;
        const float sx = inv * XS_SCALE_X;              // the conv's activation planes carry XS_SCALE_X
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            unsigned w0[2], w1[2];
            split_pair(oacc[ct][0] * sx, oacc[ct][1] * sx, w0[0], w1[0]);
            split_pair(oacc[ct][2] * sx, oacc[ct][3] * sx, w0[1], w1[1]);
            const long long c8 = h * (D / 8) + ct * 2 + (g >> 1);
            unsigned char* o = ob + ((c8 * NPL) * p.x3_tp + (t + X3_HALO)) * 16 + (g & 1) * 8;
            *reinterpret_cast<uint2*>(o) = make_uint2(w0[0], w0[1]);
            *reinterpret_cast<uint2*>(o + (long long)p.x3_tp * 16) = make_uint2(w1[0], w1[1]);
        }
        return;
    }
    float* ob = p.out + (long long)b * p.o_bs + (long long)(h * D) * p.o_cs;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(long long)(ct * 16 + 4 * g + r) * p.o_cs + t] = oacc[ct][r] * inv;
}
}  // namespace

void launch_flash_attention_x3(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out && !p.planes, "attention_x3 covers head dim 48 with the T5 bias, fp32 operands");
    constexpr size_t lds = NBUF * BUF_BYTES + sizeof(float) * (2 * BIAS_CLIP + 1);
    lds_optin(reinterpret_cast<const void*>(flash_attn_x3_kernel), (int)lds);
    const dim3 grid(cdiv(p.T, QPB) * p.H * p.B);
    hipLaunchKernelGGL(flash_attn_x3_kernel, grid, dim3(NW * 64), lds, stream, p);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
