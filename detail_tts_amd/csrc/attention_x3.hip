// Split-precision (3 x bf16) flash attention for the diffusion trunk (head dim 48, T5 relative-position bias) on gfx950.
//
// Same algorithm and register mapping as flash_attn_kernel (attention.hip): the score tile is computed transposed,
// S^T[key, query] = K^T Q, so each query's scores sit in registers of 4 lanes and P^T is already the B operand of
// O[c, query] += V[c, key] P^T[key, query].  The matrix products run on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into
// three bf16 planes (a = a0 + a1 + a2, 24 bits) and the six cross products a_i b_j (i + j <= 2) accumulated in fp32 - the result
// matches the fp32-MFMA kernel to fp32 rounding at 2.3x less matrix-pipe time (336 vs 768 SIMD-cycles per 16 x 16 score tile).
//
//   QK^T : K = 48 channels = one full 32-channel MFMA + one half-used (channels 48..63 of Q are zero, the K lanes of that half
//          re-read valid chunks).  K tile in LDS: chunks (8 channels x 16 B) [plane][c8 0..5][key 0..63]   -> linear ds_read_b128
//   P V  : one MFMA consumes 32 keys = two 16-key score tiles; k-slot 8g + r <-> key 4g + r (first tile), 8g + 4 + r <-> key
//          16 + 4g + r (second).  V tile in LDS: chunks (8 keys in that slot order) [plane][ct][u][g][channel 0..15]
// K, V and Q are split on the fly while staging (hardware v_cvt_pk_bf16_f32); P is split in registers after the softmax.
#include "attention.h"
#include "conv_x3.h"
#include "split3.h"
#include <cstdlib>

namespace dtts {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int D = 48, KT = 64, BIAS_CLIP = 64;
constexpr int KCH = 6 * KT;                    // K chunks per plane
constexpr int VCH = 3 * 2 * 4 * 16;            // V chunks per plane
constexpr int BUF_BYTES = 3 * (KCH + VCH) * 16;   // 36 KiB per stage

__device__ __forceinline__ bf16x8 as_bf(const uint4& q) { return __builtin_bit_cast(bf16x8, q); }

// the six significant cross products, smallest first
#define DTTS_X3_MFMA(acc, A, Bq)                                                          \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], Bq[0], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], Bq[1], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], Bq[2], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], Bq[0], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], Bq[1], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], Bq[0], acc, 0, 0, 0);

// K/V of one (sample, head, 64-key tile) -> the tile image the attention kernel copies into LDS.  384 threads: thread i owns K chunk
// i = (c8, key) (8 channels of one key, loads coalesced along the keys) and V chunk i = ((ct*2 + u)*4 + g)*16 + c16 (keys
// {4g..4g+3, 16+4g..+3} + 32u of channel ct*16 + c16).  Keys >= len: K clamped (finite, masked later), V zero.
__global__ __launch_bounds__(384) void kv_split_kernel(const AttnParams p) {
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int len = p.lens ? p.lens[b] : p.T;
    const int ntiles_alloc = (p.T + KT - 1) / KT;
    uint4* img = static_cast<uint4*>(p.kv3) + ((long long)(b * p.H + h) * ntiles_alloc + kt) * (BUF_BYTES / 16);
    const int s0 = kt * KT;
    if (s0 >= len) return;                         // tiles beyond the sample are never read
    const float* base = p.qkv + (long long)b * p.bs;
    const float* kp = base + (long long)(p.k_off + h * p.head_stride) * p.cs;
    const float* vp = base + (long long)(p.v_off + h * p.head_stride) * p.cs;
    float kr[8], vr[8];
    {
        const int key = tid & 63, c8 = tid >> 6;
        const int s = s0 + key, sc = s < len ? s : len - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) kr[e] = kp[(long long)(c8 * 8 + e) * p.cs + sc];
    }
    {
        const int vc = ((tid >> 7) << 4) + (tid & 15);
        const int vkey = ((tid >> 6) & 1) * 32 + ((tid >> 4) & 3) * 4;
        const bool vec_ok = ((p.cs & 3) == 0) && ((reinterpret_cast<unsigned long long>(vp) & 15ull) == 0);
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
                vr[e] = ok ? a : 0.f;
            }
        }
    }
    uint4 w0, w1, w2;
    split8(kr, w0, w1, w2);
    img[0 * KCH + tid] = w0; img[1 * KCH + tid] = w1; img[2 * KCH + tid] = w2;
    split8(vr, w0, w1, w2);
    img[3 * KCH + 0 * VCH + tid] = w0; img[3 * KCH + 1 * VCH + tid] = w1; img[3 * KCH + 2 * VCH + tid] = w2;
}

// NW waves x QT query tiles of 16 per wave.  <8, 1>: 512 threads, <= 128 VGPRs -> 16 waves per CU; <4, 2>: the K/V fragments of a
// tile are reused by two query tiles (half the LDS reads per MFMA, two independent accumulator chains per fragment).
template <int NW, int QT>
__global__ __launch_bounds__(NW * 64, (NW == 8 && QT == 1) ? 4 : 2) void flash_attn_x3_kernel(const AttnParams p) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int QPW = 16 * QT, QPB = NW * QPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_s = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);      // [129], pre-multiplied by log2(e)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nqb = (p.T + QPB - 1) / QPB;
    const int Lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = Lid % nqb, hb = Lid / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = qb * QPB;
    if (q0 >= len) return;

    const float* qp = p.qkv + (long long)b * p.bs + (long long)(p.q_off + h * p.head_stride) * p.cs;
    if (tid < 2 * BIAS_CLIP + 1) bias_s[tid] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + tid] * LOG2E;

    // ---- staging: the K/V tile images were written by kv_split_kernel in exactly the LDS layout (36 KiB per 64-key tile: K chunks
    // [plane][c8 0..5][key], V chunks [plane][ct][u][g][channel 0..15]); a tile is copied by 36 LDS-DMA pieces of 1 KiB, wave w
    // takes pieces w, w + NW, ...: no registers, no VALU
    const int ntiles = (len + KT - 1) / KT, ntiles_alloc = (p.T + KT - 1) / KT;
    const uint4* img = static_cast<const uint4*>(p.kv3) + ((long long)(b * p.H + h) * ntiles_alloc) * (BUF_BYTES / 16) + lane;
    auto fetch_tile = [&](int kt, int buf) {
        const uint4* gsrc = img + (long long)kt * (BUF_BYTES / 16);
#pragma unroll
        for (int i = 0; i < (36 + NW - 1) / NW; ++i) {
            const int pc = wave + i * NW;
            if (pc < 36)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + pc * 64),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * BUF_BYTES + pc * 1024), 16, 0, 0);
        }
    };
    fetch_tile(0, 0);

    // ---- Q fragments: B operand, lane (query j, g) holds channels kb*32 + 8g .. +7 of each plane, pre-scaled by scale*log2(e)
    const int tq0 = q0 + wave * QPW;
    const float qs = p.scale * LOG2E;
    bf16x8 qf[QT][2][3];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = tq0 + qt * 16 + j;
        const int tc = t < len ? t : len - 1;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float v[8];
            const int c0 = kb * 32 + 8 * g;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 < D) ? qp[(long long)(c0 + e < D ? c0 + e : 0) * p.cs + tc] * qs : 0.f;
            uint4 w0, w1, w2;
            split8(v, w0, w1, w2);
            qf[qt][kb][0] = as_bf(w0);
            qf[qt][kb][1] = as_bf(w1);
            qf[qt][kb][2] = as_bf(w2);
        }
    }

    floatx4 oacc[3][QT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) oacc[ct][qt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const bool wave_active = tq0 < len;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // A-operand chunk column of this lane for the two channel blocks: kb 0 -> c8 = g; kb 1 -> c8 = 4 + (g & 1) (lanes g >= 2 meet
    // zero Q channels 48..63, any finite chunk will do)
    const int kcol0 = g * KT + j, kcol1 = (4 + (g & 1)) * KT + j;

    for (int kt = 0; kt < ntiles; ++kt) {
        const int s0 = kt * KT, buf = kt & 1;
        if (kt + 1 < ntiles) fetch_tile(kt + 1, buf ^ 1);
        if (wave_active) {
            const uint4* Kb = reinterpret_cast<const uint4*>(smem + buf * BUF_BYTES);
            const uint4* Vb = Kb + 3 * KCH;
            const bool full_tile = (s0 + KT <= len);
            // every (key, query) pair of this wave's tile beyond the bias window on one side -> one bucket, no table look-ups
            const bool far_hi = s0 - (tq0 + QPW - 1) >= BIAS_CLIP, far_lo = (s0 + KT - 1) - tq0 <= -BIAS_CLIP;
            const bool far = (far_hi || far_lo) && full_tile;
            const float bfar = bias_s[far_hi ? 2 * BIAS_CLIP : 0];
            // ---- S^T = K^T Q
            floatx4 sacc[QT][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) sacc[qt][ks] = floatx4{0.f, 0.f, 0.f, 0.f};
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = as_bf(Kb[pl * KCH + kcol0 + ks * 16]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) { DTTS_X3_MFMA(sacc[qt][ks], a, qf[qt][0]) }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = as_bf(Kb[pl * KCH + kcol1 + ks * 16]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) { DTTS_X3_MFMA(sacc[qt][ks], a, qf[qt][1]) }
            }
            // ---- per 32-key half u: bias, length mask, online softmax in the log2 domain, P split, O += V P^T
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 pf[QT][3];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int t = tq0 + qt * 16 + j;
                    float mx = -INFINITY;
                    if (far) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                sacc[qt][2 * u + q][r] += bfar;
                                mx = fmaxf(mx, sacc[qt][2 * u + q][r]);
                            }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int sk = s0 + (2 * u + q) * 16 + 4 * g + r;
                                float v = sacc[qt][2 * u + q][r];
                                int off = sk - t;
                                off = off < -BIAS_CLIP ? -BIAS_CLIP : (off > BIAS_CLIP ? BIAS_CLIP : off);
                                v += bias_s[off + BIAS_CLIP];
                                if (!full_tile) v = (sk >= len) ? -INFINITY : v;
                                sacc[qt][2 * u + q][r] = v;
                                mx = fmaxf(mx, v);
                            }
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float m_new = fmaxf(m_run[qt], mx);
                    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_use);
                    float sum = 0.f;
                    float pv[8];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(sacc[qt][2 * u + q][r] - m_use);
                            pv[q * 4 + r] = e;
                            sum += e;
                        }
                    sum += __shfl_xor(sum, 16);
                    sum += __shfl_xor(sum, 32);
                    l_run[qt] = l_run[qt] * alpha + sum;
                    m_run[qt] = m_new;
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) oacc[ct][qt] *= alpha;
                    uint4 w0, w1, w2;
                    split8(pv, w0, w1, w2);
                    pf[qt][0] = as_bf(w0);
                    pf[qt][1] = as_bf(w1);
                    pf[qt][2] = as_bf(w2);
                }
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    bf16x8 a[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[pl] = as_bf(Vb[pl * VCH + ((ct * 2 + u) * 4 + g) * 16 + j]);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) { DTTS_X3_MFMA(oacc[ct][qt], a, pf[qt]) }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (!wave_active) return;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int t = tq0 + qt * 16 + j;
        if (t >= len) continue;
        const float inv = 1.f / l_run[qt];
        if (p.out_x3) {
            // lane (j, g) holds channels ct*16 + 4g + r of query t: half (g & 1) of the 8-channel chunk c8 = h*6 + ct*2 + (g >> 1)
            unsigned char* ob = static_cast<unsigned char*>(p.out_x3) + ((long long)b * (p.H * D / 8) * 3) * p.x3_tp * 16;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) {
                unsigned w0[2], w1[2], w2[2];
                split_pair(oacc[ct][qt][0] * inv, oacc[ct][qt][1] * inv, w0[0], w1[0], w2[0]);
                split_pair(oacc[ct][qt][2] * inv, oacc[ct][qt][3] * inv, w0[1], w1[1], w2[1]);
                const long long c8 = h * (D / 8) + ct * 2 + (g >> 1);
                unsigned char* o = ob + ((c8 * 3) * p.x3_tp + (t + X3_HALO)) * 16 + (g & 1) * 8;
                *reinterpret_cast<uint2*>(o) = make_uint2(w0[0], w0[1]);
                *reinterpret_cast<uint2*>(o + (long long)p.x3_tp * 16) = make_uint2(w1[0], w1[1]);
                *reinterpret_cast<uint2*>(o + 2LL * p.x3_tp * 16) = make_uint2(w2[0], w2[1]);
            }
        } else {
            float* ob = p.out + (long long)b * p.o_bs + (long long)(h * D) * p.o_cs;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(long long)(ct * 16 + 4 * g + r) * p.o_cs + t] = oacc[ct][qt][r] * inv;
        }
    }
}
}  // namespace

size_t attn_x3_kv_bytes(int B, int H, int T) { return (size_t)B * H * cdiv(T, KT) * BUF_BYTES; }

template <int NW, int QT>
static void launch_x3(const AttnParams& p, hipStream_t stream) {
    constexpr size_t lds = 2 * BUF_BYTES + sizeof(float) * (2 * BIAS_CLIP + 1);
    static bool attr = false;
    if (!attr) {
        DTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_x3_kernel<NW, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((flash_attn_x3_kernel<NW, QT>), dim3(cdiv(p.T, NW * QT * 16) * p.H * p.B), dim3(NW * 64), lds, stream, p);
}

void launch_flash_attention_x3(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out, "attention_x3 covers head dim 48 with the T5 bias only");
    DTTS_REQUIRE(p.kv3, "attention_x3 needs the K/V tile-image scratch (attn_x3_kv_bytes)");
    static const int variant = []() { const char* v = getenv("DTTS_ATTN_X3_VARIANT"); return v ? atoi(v) : 0; }();
    hipLaunchKernelGGL(kv_split_kernel, dim3(cdiv(p.T, KT), p.H, p.B), dim3(384), 0, stream, p);
    if (variant == 1) launch_x3<4, 2>(p, stream);
    else if (variant == 2) launch_x3<8, 2>(p, stream);
    else launch_x3<8, 1>(p, stream);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
