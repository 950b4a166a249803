// Split-precision flash attention of the diffusion trunk on the 32 x 32 matrix instruction (head dim 48, T5 relative-position bias,
// operands = the AttnPlanes images the qkv conv wrote; vqvae/utils/diff_util.py:146-169 AttentionBlock / QKVAttentionLegacy).
//
// v_mfma_f32_16x16x32_f16 sustains only half the flops per cycle of v_mfma_f32_32x32x16_f16 on gfx950 (tools/ubench/mfma_peak16.hip:
// 1.3 vs 2.5 PFLOP/s), so a wave owns 32 queries instead of 16 and both products run on the 32 x 32 shape:
//   S^T[key 32, query 32] += K^T[key, c 16] Q[c 16, query]      3 channel steps (D = 48 = 3 x 16: no half-used instruction),
//   O[c 32, query 32]     += V[c, key 16] P^T[key 16, query]    2 channel tiles (48 of 64 rows used), 2 key steps per 32-key block.
// The score tile comes out transposed, so a lane (query q = lane & 31, half hh = lane >> 5) holds 16 keys of its query per 32-key
// block - rows (r & 3) + 8 (r >> 2) + 4 hh of the MFMA C layout - and those registers ARE the B operand of the PV product once the V
// chunks are stored in the matching key order: k-slot 8 hh + e of key step j  <->  key 16 j + (e & 3) + 8 (e >> 2) + 4 hh.
// Softmax: one cross-lane exchange (the query's two lanes), lazy running maximum decided per query (attention_x3.hip), denominator
// accumulated per lane.  Every fp32 operand is two scaled fp16 planes, three cross products per product (split3.h, conv_x3.h).
// 256 threads = 4 waves x 32 queries share each 64-key K/V tile (24 KiB image, moved by LDS-DMA; K and V double-buffered with half a
// tile of skew so that a wave's QK^T of the next tile overlaps its softmax of the current one).
#include "attention.h"
#include "conv_x3.h"
#include "split3.h"

namespace dtts {

namespace {
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int D = 48, KT = 64, NW = 4, QPW = 32, QPB = NW * QPW, BIAS_CLIP = 64;
constexpr int KCH = 6 * KT;                    // K chunks per plane and tile
constexpr int VCH = 8 * D;                     // V chunks per plane and tile: (key block 2, key step 2, half 2) x channel 48
constexpr int NPL = XS_PLANES;
constexpr int BUF_BYTES = NPL * (KCH + VCH) * 16;   // one tile image
constexpr float QK_SCALE = 16.f, P_SHIFT = 10.f, V_SCALE = 16.f, M_SLACK = 3.f;
static_assert(BUF_BYTES == AttnPlanes::TILE_BYTES, "the K/V tile image is the LDS stage image");

__device__ __forceinline__ hf8 as_hf(const uint4& q) { return __builtin_bit_cast(hf8, q); }

// the three significant cross products, smallest first
#define DTTS_X3W_MFMA(acc, A, Bq)                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], Bq[0], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], Bq[1], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], Bq[0], acc, 0, 0, 0);

__global__ __launch_bounds__(256, 3) void flash_attn_x3w_kernel(const AttnParams p) {
    constexpr float LOG2E = 1.4426950408889634f;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_s = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);        // [129], pre-multiplied by log2(e)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 31, hh = lane >> 5;
    const int nqb = (p.T + QPB - 1) / QPB;
    const int Lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = Lid % nqb, hb = Lid / nqb;
    const int h = hb % p.H, b = hb / p.H;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = qb * QPB;
    if (q0 >= len) return;
    const int Tq = AttnPlanes::tq(p.T);
    const unsigned char* himg = static_cast<const unsigned char*>(p.planes) + ((size_t)b * p.H + h) * AttnPlanes::head_bytes(p.T);
    const unsigned char* kvimg = himg + AttnPlanes::q_bytes(p.T);
    if (tid < 2 * BIAS_CLIP + 1) bias_s[tid] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + tid] * LOG2E;

    // ---- Q fragments (B operand): lane (query q, half hh) holds channels 16 s + 8 hh .. + 7 = chunk 2 s + hh, both planes
    const int tq0 = q0 + wave * QPW, t = tq0 + q;
    hf8 qf[3][NPL];
    {
        const int tc = t < Tq ? t : Tq - 1;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                qf[s][pl] = as_hf(*reinterpret_cast<const uint4*>(himg + ((size_t)(pl * (D / 8) + 2 * s + hh) * Tq + tc) * 16));
    }
    f16v oacc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.f;
    float m_run = -INFINITY, l_lane = 0.f;
    const int ntiles = (len + KT - 1) / KT;
    const bool wave_active = tq0 < len;

    // LDS-DMA: a tile image is 12 K pieces + 12 V pieces of 1 KiB; three of each per wave.  K and V are staged with HALF A TILE OF SKEW:
    // during iteration kt the LDS holds K(kt+1) and V(kt) - the wave computes the score tile of the NEXT key tile (matrix pipe) while it
    // runs the softmax of the current one (vector pipe), the two being independent - and the DMA brings K(kt+2) and V(kt+1).
    constexpr int KBYTES = NPL * KCH * 16, VBYTES = NPL * VCH * 16;      // 12 KiB each; LDS: K stage 0 | K stage 1 | V stage 0 | V stage 1
    auto dma_k = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + NW * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kvimg + (size_t)kt * AttnPlanes::TILE_BYTES + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (kt & 1) * KBYTES + piece * 1024), 16, 0, 0);
        }
    };
    auto dma_v = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + NW * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kvimg + (size_t)kt * AttnPlanes::TILE_BYTES + KBYTES + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + 2 * KBYTES + (kt & 1) * VBYTES + piece * 1024), 16, 0, 0);
        }
    };
    // score tile of key tile kt: 2 key blocks x 3 channel steps x 3 products
    auto qk_tile = [&](int kt, f16v* sacc) {
        const uint4* Kb = reinterpret_cast<const uint4*>(smem + (kt & 1) * KBYTES);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                hf8 a[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Kb[pl * KCH + (2 * s + hh) * KT + kb * 32 + q]);
                DTTS_X3W_MFMA(sacc[kb], a, qf[s])
            }
        }
    };
    dma_k(0);
    dma_v(0);
    if (ntiles > 1) dma_k(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // V chunk of this lane for the two channel tiles: channel ct 32 + q; the padding rows 48..63 of tile 1 re-read valid chunks
    // (their accumulator rows are never stored)
    const int vch0 = q, vch1 = q < 16 ? 32 + q : 16 + q;
    f16v sacc[2];
    if (wave_active) qk_tile(0, sacc);
    __syncthreads();                                 // K(0)'s stage is refilled by the first iteration

    for (int kt = 0; kt < ntiles; ++kt) {
        const int s0 = kt * KT;
        if (kt + 2 < ntiles) dma_k(kt + 2);          // into the stage K(kt) left: every wave finished QK(kt) before the last barrier
        if (kt + 1 < ntiles) dma_v(kt + 1);          // into the stage V(kt-1) left
        if (wave_active) {
            f16v snext[2];
            if (kt + 1 < ntiles) qk_tile(kt + 1, snext);
            const uint4* Vb = reinterpret_cast<const uint4*>(smem + 2 * KBYTES + (kt & 1) * VBYTES);
            const bool full_tile = (s0 + KT <= len);
            // every (key, query) pair of this wave's tile beyond the bias window on one side -> one bucket, no table look-ups
            const bool far_hi = s0 - (tq0 + QPW - 1) >= BIAS_CLIP, far_lo = (s0 + KT - 1) - tq0 <= -BIAS_CLIP;
            const bool far = (far_hi || far_lo) && full_tile;
            const float bfar = bias_s[far_hi ? 2 * BIAS_CLIP : 0];
            constexpr float SU = 1.f / (QK_SCALE * QK_SCALE);            // the score accumulator holds 256 S
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                // exponent arguments e = s + bias (log2 domain) of this lane's 16 keys of the block, and their maximum
                float mx = -INFINITY;
                if (far) {
                    float r0 = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) r0 = fmaxf(r0, sacc[kb][r]);
                    mx = fmaf(r0, SU, bfar);                             // SU > 0: the maximum commutes with the affine map
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int s = s0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        int off = s - t;
                        off = off < -BIAS_CLIP ? -BIAS_CLIP : (off > BIAS_CLIP ? BIAS_CLIP : off);
                        float v = fmaf(sacc[kb][r], SU, bias_s[off + BIAS_CLIP]);
                        if (!full_tile) v = (s >= len) ? -INFINITY : v;
                        sacc[kb][r] = v;
                        mx = fmaxf(mx, v);
                    }
                }
                // lazy running maximum, decided per query (see attention_x3.hip): P = exp2(e - m + 10) <= 8192 otherwise
                if (__builtin_amdgcn_ballot_w64(mx > m_run + M_SLACK) != 0ull) {
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const float m_new = (mx > m_run + M_SLACK) ? mx : m_run;
                    const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                    l_lane *= alpha;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) oacc[ct] *= alpha;
                    m_run = m_new;
                }
                const float m_sub = ((m_run == -INFINITY) ? 0.f : m_run) - P_SHIFT;
                // Regular VALU work is NOT hidden under the matrix pipe on this chip (tools/ubench/mfma_peak16.hip: the two add up), only
                // transcendentals are: the affine maps and the sums run as packed 2 x fp32 instructions (vector arithmetic below).
                f16v ev;                                                 // exponent arguments minus the running maximum
                if (far) ev = sacc[kb] * SU + (bfar - m_sub);            // 1024 P: the scale is free in the exponent
                else ev = sacc[kb] - m_sub;
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(ev[r]);
                {
                    typedef float f8v __attribute__((ext_vector_type(8)));
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    const f8v a8 = f8v{pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7]} + f8v{pv[8], pv[9], pv[10], pv[11], pv[12], pv[13], pv[14], pv[15]};
                    const f4v a4 = __builtin_shufflevector(a8, a8, 0, 1, 2, 3) + __builtin_shufflevector(a8, a8, 4, 5, 6, 7);
                    const f2v a2 = __builtin_shufflevector(a4, a4, 0, 1) + __builtin_shufflevector(a4, a4, 2, 3);
                    l_lane += a2[0] + a2[1];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {                            // key step j: this lane's registers 8 j .. 8 j + 7
                    hf8 pf[NPL];
                    {
                        uint4 w0, w1;
                        split8_inrange(pv + 8 * j, w0, w1);
                        pf[0] = as_hf(w0);
                        pf[1] = as_hf(w1);
                    }
                    const int vrow = ((kb * 2 + j) * 2 + hh) * D;
                    hf8 a[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Vb[pl * VCH + vrow + vch0]);
                    DTTS_X3W_MFMA(oacc[0], a, pf)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) a[pl] = as_hf(Vb[pl * VCH + vrow + vch1]);
                    DTTS_X3W_MFMA(oacc[1], a, pf)
                }
            }
            sacc[0] = snext[0];
            sacc[1] = snext[1];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // K(kt+2), V(kt+1) have landed (this wave's pieces; the barrier covers the others')
        __syncthreads();
    }

    if (!wave_active) return;
    float l_run = l_lane;
    l_run += __shfl_xor(l_run, 32);
    if (t >= len) return;
    const float inv = 1.f / (l_run * V_SCALE);          // acc = (1024 P)(16 V), l_run = sum of 1024 P
    // lane (q, hh) holds channels ct 32 + 8 rg + 4 hh + (0..3) of query t: half hh of the 8-channel chunk ct 4 + rg
    if (p.out_x3) {
        unsigned char* ob = static_cast<unsigned char*>(p.out_x3) + ((long long)b * (p.H * D / 8) * NPL) * p.x3_tp * 16;
        const float sx = inv * XS_SCALE_X;              // the conv's activation planes carry XS_SCALE_X
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rg = 0; rg < (ct ? 2 : 4); ++rg) {
                unsigned w0[2], w1[2];
                split_pair(oacc[ct][4 * rg] * sx, oacc[ct][4 * rg + 1] * sx, w0[0], w1[0]);
                split_pair(oacc[ct][4 * rg + 2] * sx, oacc[ct][4 * rg + 3] * sx, w0[1], w1[1]);
                const long long c8 = h * (D / 8) + ct * 4 + rg;
                unsigned char* o = ob + ((c8 * NPL) * p.x3_tp + (t + X3_HALO)) * 16 + hh * 8;
                *reinterpret_cast<uint2*>(o) = make_uint2(w0[0], w0[1]);
                *reinterpret_cast<uint2*>(o + (long long)p.x3_tp * 16) = make_uint2(w1[0], w1[1]);
            }
        return;
    }
    float* ob = p.out + (long long)b * p.o_bs + (long long)(h * D) * p.o_cs;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int rg = 0; rg < (ct ? 2 : 4); ++rg)
#pragma unroll
            for (int e = 0; e < 4; ++e) ob[(long long)(ct * 32 + 8 * rg + 4 * hh + e) * p.o_cs + t] = oacc[ct][4 * rg + e] * inv;
}
}  // namespace

// operands = AttnPlanes images (p.planes)
void launch_flash_attention_x3w(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out && p.planes, "attention_x3w covers head dim 48 with the T5 bias on operand images");
    constexpr size_t lds = 2 * BUF_BYTES + sizeof(float) * (2 * BIAS_CLIP + 1);
    lds_optin(reinterpret_cast<const void*>(flash_attn_x3w_kernel), (int)lds);
    const dim3 grid(cdiv(p.T, QPB) * p.H * p.B);
    hipLaunchKernelGGL(flash_attn_x3w_kernel, grid, dim3(NW * 64), lds, stream, p);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
