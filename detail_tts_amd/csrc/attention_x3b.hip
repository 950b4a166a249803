// hipcc-flags: -fno-honor-nans
// (build.py reads the line above: no NaN canonicalisation in front of every v_max3_f32; infinities keep their meaning.)
// Split-precision flash attention of the diffusion trunk, software-pipelined over 32-key blocks (head dim 48, T5 relative-position bias;
// operands = the AttnPlanes images the qkv conv wrote; vqvae/utils/diff_util.py:136-215 AttentionBlock / QKVAttentionLegacy,
// vqvae/utils/xtransformers.py:146-186 RelativePositionBias).
//
// Arithmetic, layouts and the lane <-> key mapping are those of attention_x3w.hip (the round 2 - 4 kernel, kept behind
// DTTS_ATTN_KERNEL=w for A/B runs):
//   S^T[key 32, query 32] += K^T[key, c 16] Q[c 16, query]      3 channel steps x 3 split products   (v_mfma_f32_32x32x16_f16)
//   O[c 32, query 32]     += V[c, key 16] P^T[key 16, query]    2 channel tiles x 2 key steps x 3 split products
// a lane (query q = lane & 31, half hh = lane >> 5) holds the 16 keys (r & 3) + 8 (r >> 2) + 4 hh of its query per 32-key block, and
// those registers are the B operand of the PV product.  What changed is the SCHEDULE and the vector-instruction count (VERDICT r04
// item 1: 312 vector instructions against 42 MFMAs per 64-key tile, matrix pipe busy 34 %):
//   * the unit of work is a 32-key BLOCK and a step is pipelined over three of them - QK^T of block b + 1, softmax of block b, PV of
//     block b - 1 - so none of a step's 21 MFMAs depends on its vector work (attn_step below);
//   * the accumulators are touched by ONE code path: every join at which they are live on both sides (block-class arms, a C++ rescale
//     branch, `if (wave active)` inside the loop) cost a copy of 32 registers per block; the block classes run as separate loops;
//   * FAR blocks (every (key, query) pair beyond the bias window on one side: one bucket) look nothing up; NEAR blocks read the bias
//     from an LDS table extended to +-128, at constant offsets from a per-lane base (no clamp, no address arithmetic per score);
//   * the softmax denominator comes out of the PV MFMAs (the 16 unused rows of the second channel tile read a "ones" fragment);
//   * the numerators are split with v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (3 instructions per pair instead of 4);
//   * K tiles are double-buffered, V travels in 32-key blocks through a ring of four 6 KiB slots one tile behind K, 24 LDS-DMA pieces
//     and one barrier per 64 keys as before.
// Vector instructions per launch 1.92e7 -> 1.51e7 (rocprofv3 SQ_INSTS_VALU, B = 8, T = 936), 102.6 -> 94.5 us alone
// (profiles/r05_attn_ablate_pipelined.txt); under the bench the chip runs at its 1400 W power limit (profiles/r05_power_bench.txt:
// 1340 - 1360 W, 1.97 GHz) and the step time does not move (DESIGN.md par. 4).
#include <atomic>
#include <type_traits>

#include "attention.h"
#include "conv_x3.h"
#include "split3.h"

namespace dtts {

namespace {
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int D = 48, KT = 64, QPW = 32, BIAS_CLIP = 64;
constexpr int KCH = 6 * KT;                    // K chunks per plane and tile
constexpr int NPL = XS_PLANES;
constexpr int KBYTES = NPL * KCH * 16;         // one K stage: 12 KiB
constexpr int VPLANE_G = 8 * D * 16;           // one V plane of a tile in the global image: 6 KiB = (u 2, j 2, hh 2) x 48 chunks
constexpr int VHALF = 4 * D * 16;              // one plane of one 32-key block: (j 2, hh 2) x 48 chunks = 3 KiB
constexpr int VSLOT = NPL * VHALF;             // ring slot: one block, both planes
constexpr int EXT_HALF = 128, EXT_N = 2 * EXT_HALF + 1;
// LDS: K stage 0 | K stage 1 | V slots 0..3 | "ones" area with the extended bias table in its first gap.  The ones area feeds the 16
// unused rows (48..63) of the PV product's second channel tile: lanes q >= 16 read their V fragments from it instead of from the slot -
// plane 0 = 16.0 (V's scale), plane 1 = 0 at the fragment offsets (plane VHALF, key step 2 D 16) - so row 48 of the accumulator IS the
// softmax denominator (sum of P, both planes, fp32 accumulate) and the 16 vector adds per block are gone.  Only those four 16-byte
// chunks of the area are ever read as fragments, so the bias table lives between the first two: 53 776 bytes, three workgroups per CU.
constexpr int ONES_BYTES = VHALF + 2 * D * 16 + 16;
constexpr int LDS_K = 0, LDS_V = 2 * KBYTES, LDS_ONES = LDS_V + 4 * VSLOT, LDS_EXT = LDS_ONES + 16, LDS_BYTES = LDS_ONES + ONES_BYTES;
static_assert(16 + EXT_N * 4 <= 2 * D * 16, "the bias table fits between the two ones chunks");
static_assert(3 * ((LDS_BYTES + 511) / 512 * 512) <= 160 * 1024, "three workgroups per CU");
constexpr float QK_SCALE = 16.f, P_SHIFT = 10.f, V_SCALE = 16.f, M_SLACK = 3.f;
constexpr float SU = 1.f / (QK_SCALE * QK_SCALE);            // the score accumulator holds 256 S
static_assert(2 * KBYTES + NPL * VPLANE_G * 2 == 2 * AttnPlanes::TILE_BYTES, "two tile images");

enum { FAR = 0, FAR_MASK = 1, NEAR = 2 };

__device__ __forceinline__ hf8 as_hf(const uint4& q) { return __builtin_bit_cast(hf8, q); }
__device__ __forceinline__ constexpr int roff(int r) { return (r & 3) + 8 * (r >> 2); }

// p, q (inside fp16's range) -> one packed word per plane: h0 = fp16(x) pair, h1 = fp16(x - h0) pair (the residual is exact in fp32;
// v_fma_mixlo/hi_f16 compute it in fp32 from the fp16 half in place and round once, like v_cvt_pk_f16_f32 of the fp32 residual)
__device__ __forceinline__ void split_pair_mix(float x, float y, unsigned& w0, unsigned& w1) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, y};
    w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, hf2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(w1) : "v"(w0), "v"(x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(w1) : "v"(w0), "v"(y));
}

#define DTTS_X3B_MFMA(acc, A, Bq)                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], Bq[0], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], Bq[1], acc, 0, 0, 0);             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], Bq[0], acc, 0, 0, 0);

struct WaveState {
    hf8 qf[3][NPL];        // Q fragments (B operand of QK^T)
    f16v oacc[2];          // O accumulators: channels 0..31 | 32..47, row 48 (oacc[1][8]) = the denominator, 15 unused rows
    f16v s[2];             // score sets: s[b & 1] holds block b
    hf8 pf[2][NPL];        // P planes of the block whose PV product is still to come (key step j: the lane's registers 8 j .. 8 j + 7)
    float m_run;
    float alpha;           // pending rescale of the accumulators (decided by the last softmax, applied before the next PV product)
    unsigned long long need;   // ... lanes mask: != 0 when some query of the wave raised its maximum
};

// One step b of a wave, software-pipelined over THREE blocks so that none of its 21 MFMAs depends on anything computed in the step:
//   QK^T of block b + 1 (9 MFMAs, into st.s[(b + 1) & 1]),  PV of block b - 1 (12 MFMAs, from st.pf = the planes the previous step made),
//   softmax of block b (vector pipe: st.s[b & 1] -> st.pf), one MFMA : three to four vector instructions (sched_group_barrier).
// Measured on the instruction mix alone (tools/ubench/attn_phase.hip): a wave that alternates QK^T -> softmax -> PV phases runs 1307
// cycles per step at three waves per SIMD, this schedule 1188, the matrix pipe alone 987 (random operands: the power-limited clock).
// The rescale a block's softmax decides (a query's maximum grew by more than 2^3) must reach the accumulators AFTER the PV product of
// the block before it and BEFORE its own: it is applied at the top of the next step - by then the MFMAs it has to wait for are a whole
// step old.  It is ONE asm statement with its own skip branch and tied operands: as a C++ branch it costs a copy of the 32 accumulator
// registers per block on the path not taken.
//   kb_addr: this lane's byte address of K chunk (plane 0, c8 = hh, key q) of the stage that holds block b + 1's tile (+ its half)
//   v_addr0 / v_addr1: byte address of V chunk (plane 0, j 0, hh, channel vch0 / vch1) of block b - 1; lanes q >= 16 pass the ones area
//   GENERAL: the block class of b is NEAR or FAR_MASK (bias from the extended table / tail mask); else FAR (one bias, no mask)
template <int PAR /* b & 1 */, bool DO_QK, bool DO_PV, bool DO_SM, bool GENERAL, int ABL>
__device__ __forceinline__ void attn_step(WaveState& st, const unsigned char* smem, unsigned kb_addr, unsigned v_addr0, unsigned v_addr1, bool near,
                                          const float* ext_lane, int lim, float bfar) {
    f16v& sq = st.s[PAR ^ 1];
    f16v& ss = st.s[PAR];
    if (DO_PV) {
        // The accumulators are pinned to v[200:231] HERE (physical-register constraints), so that the 32 multiplies can name their
        // registers: with one tied operand per element the allocator scattered the two tuples over single registers and re-assembled
        // them with ~40 moves in front of every group of MFMAs.  The copies to / from the pinned registers coalesce away when the
        // allocator keeps the tuples there for the whole kernel, which it does (no v_mov in the loop: checked in the ISA).
        asm volatile(
            "s_cmp_eq_u64 %[need], 0\n\t"
            "s_cbranch_scc1 .Lx3b_skip_%=\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"       // an MFMA result needs 19 wait states before a VALU read (the hazard recognizer does not see asm)
            "v_mul_f32 v200, %[al], v200\n\t"
            "v_mul_f32 v201, %[al], v201\n\t"
            "v_mul_f32 v202, %[al], v202\n\t"
            "v_mul_f32 v203, %[al], v203\n\t"
            "v_mul_f32 v204, %[al], v204\n\t"
            "v_mul_f32 v205, %[al], v205\n\t"
            "v_mul_f32 v206, %[al], v206\n\t"
            "v_mul_f32 v207, %[al], v207\n\t"
            "v_mul_f32 v208, %[al], v208\n\t"
            "v_mul_f32 v209, %[al], v209\n\t"
            "v_mul_f32 v210, %[al], v210\n\t"
            "v_mul_f32 v211, %[al], v211\n\t"
            "v_mul_f32 v212, %[al], v212\n\t"
            "v_mul_f32 v213, %[al], v213\n\t"
            "v_mul_f32 v214, %[al], v214\n\t"
            "v_mul_f32 v215, %[al], v215\n\t"
            "v_mul_f32 v216, %[al], v216\n\t"
            "v_mul_f32 v217, %[al], v217\n\t"
            "v_mul_f32 v218, %[al], v218\n\t"
            "v_mul_f32 v219, %[al], v219\n\t"
            "v_mul_f32 v220, %[al], v220\n\t"
            "v_mul_f32 v221, %[al], v221\n\t"
            "v_mul_f32 v222, %[al], v222\n\t"
            "v_mul_f32 v223, %[al], v223\n\t"
            "v_mul_f32 v224, %[al], v224\n\t"
            "v_mul_f32 v225, %[al], v225\n\t"
            "v_mul_f32 v226, %[al], v226\n\t"
            "v_mul_f32 v227, %[al], v227\n\t"
            "v_mul_f32 v228, %[al], v228\n\t"
            "v_mul_f32 v229, %[al], v229\n\t"
            "v_mul_f32 v230, %[al], v230\n\t"
            "v_mul_f32 v231, %[al], v231\n\t"
            "s_nop 3\n"
            ".Lx3b_skip_%=:"
            : "+{v[200:215]}"(st.oacc[0]), "+{v[216:231]}"(st.oacc[1])
            : [al] "v"(st.alpha), [need] "s"(st.need)
            : "scc");
    }
    // ---- LDS fragments, then the 21 MFMAs: they depend on registers of earlier steps only
    hf8 ka[3][NPL], va[2][2][NPL];                           // K fragments of the 3 channel steps; V fragments [key step][channel tile]
    if (DO_QK && !(ABL & 32)) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ka[s][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + kb_addr + pl * (KCH * 16) + s * (2 * KT * 16)));
    }
    if (DO_PV && !(ABL & 32)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                va[j][0][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + v_addr0 + pl * VHALF + j * (2 * D * 16)));
                va[j][1][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + v_addr1 + pl * VHALF + j * (2 * D * 16)));
            }
    }
    if (ABL & 32) {          // measurement build: no fragment reads
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ka[s][pl] = st.qf[s][pl];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) va[j][0][pl] = va[j][1][pl] = st.qf[j][pl];
    }
    if (DO_QK && !(ABL & 16)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[r] = 0.f;
        DTTS_X3B_MFMA(sq, ka[0], st.qf[0])
        DTTS_X3B_MFMA(sq, ka[1], st.qf[1])
        DTTS_X3B_MFMA(sq, ka[2], st.qf[2])
    }
    if (DO_PV && !(ABL & 8)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            DTTS_X3B_MFMA(st.oacc[0], va[j][0], st.pf[j])
            DTTS_X3B_MFMA(st.oacc[1], va[j][1], st.pf[j])
        }
    }
    // ---- softmax of block b: exponent arguments e = s + bias (log2 domain) of this lane's 16 keys and their maximum, the lazy running
    // maximum (decided per query: both lanes of a query see the pair's maximum; P = exp2(e - m + 10) <= 8192 otherwise), the P planes
    if (DO_SM && !(ABL & 4)) {
        f16v e;                                              // GENERAL only
        float mx;
        if (GENERAL) {
            if (near) {              // bias from the extended table
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaf(ss[r], SU, ext_lane[roff(r)]);
                    e[r] = (roff(r) < lim) ? v : -INFINITY;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = (roff(r) < lim) ? fmaf(ss[r], SU, bfar) : -INFINITY;
            }
            mx = fmaxf(fmaxf(e[0], e[1]), e[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, e[r]), e[r + 1]);
            mx = fmaxf(mx, e[15]);
        } else {
            float r0 = fmaxf(fmaxf(ss[0], ss[1]), ss[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) r0 = fmaxf(fmaxf(r0, ss[r]), ss[r + 1]);
            r0 = fmaxf(r0, ss[15]);
            mx = fmaf(r0, SU, bfar);                         // SU > 0: the maximum commutes with the affine map
        }
        {
            // need != 0: the partner lane's maximum (v_permlane32_swap), m_new = the pair's maximum where it exceeds m_run + 2^3,
            // alpha = exp2(m_run - m_new) (0 while m_run = -inf); the multiplication of the accumulators is the NEXT step's first act
            st.need = __builtin_amdgcn_ballot_w64(mx > st.m_run + M_SLACK);
            float tb;
            asm volatile(
                "s_cmp_eq_u64 %[need], 0\n\t"
                "s_cbranch_scc1 .Lx3b_skipm_%=\n\t"
                "v_mov_b32 %[ta], %[mx]\n\t"
                "v_mov_b32 %[tb], %[mx]\n\t"
                "s_nop 1\n\t"
                "v_permlane32_swap_b32 %[ta], %[tb]\n\t"              // ta = mx[lane & 31], tb = mx[32 + (lane & 31)] in both halves
                "s_nop 1\n\t"
                "v_max_f32 %[ta], %[ta], %[tb]\n\t"                   // the pair's maximum
                "v_add_f32 %[tb], 0x40400000, %[m]\n\t"              // m_run + M_SLACK
                "v_cmp_gt_f32 vcc, %[ta], %[tb]\n\t"
                "v_cndmask_b32 %[tb], %[m], %[ta], vcc\n\t"           // m_new
                "v_cmp_neq_f32 vcc, 0xff800000, %[tb]\n\t"
                "v_cndmask_b32 %[ta], 0, %[tb], vcc\n\t"              // m_new, or 0 while it is -inf
                "v_sub_f32 %[ta], %[m], %[ta]\n\t"
                "v_exp_f32 %[ta], %[ta]\n\t"                          // alpha
                "v_mov_b32 %[m], %[tb]\n\t"
                "s_nop 0\n"
                ".Lx3b_skipm_%=:"
                : [m] "+v"(st.m_run), [ta] "=&v"(st.alpha), [tb] "=&v"(tb)
                : [mx] "v"(mx), [need] "s"(st.need)
                : "vcc", "scc");
        }
        const float m_sub = ((st.m_run == -INFINITY) ? 0.f : st.m_run) - P_SHIFT;
        const float c0 = bfar - m_sub;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                pv[k] = __builtin_amdgcn_exp2f(GENERAL ? e[8 * j + k] - m_sub : fmaf(ss[8 * j + k], SU, c0));      // 1024 P: the scale is free in the exponent
            uint4 w0, w1;
            split_pair_mix(pv[0], pv[1], w0.x, w1.x);
            split_pair_mix(pv[2], pv[3], w0.y, w1.y);
            split_pair_mix(pv[4], pv[5], w0.z, w1.z);
            split_pair_mix(pv[6], pv[7], w0.w, w1.w);
            st.pf[j][0] = as_hf(w0);
            st.pf[j][1] = as_hf(w1);
        }
    }
    if (DO_SM && (ABL & 4)) st.need = 0;
    if (DO_QK && DO_PV && DO_SM && !GENERAL && !(ABL & 28)) {
        // the hot instantiation: one MFMA, then three or four vector / transcendental instructions (21 MFMAs : ~72)
        constexpr int M_MFMA = 0x8, M_VALU = 0x2 | 0x400;
#define DTTS_X3B_SGB2                                          \
    __builtin_amdgcn_sched_group_barrier(M_MFMA, 1, 0);       \
    __builtin_amdgcn_sched_group_barrier(M_VALU, 3, 0);       \
    __builtin_amdgcn_sched_group_barrier(M_MFMA, 1, 0);       \
    __builtin_amdgcn_sched_group_barrier(M_VALU, 4, 0);
        DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2 DTTS_X3B_SGB2
        __builtin_amdgcn_sched_group_barrier(M_MFMA, 1, 0);
        __builtin_amdgcn_sched_group_barrier(M_VALU, 4, 0);
#undef DTTS_X3B_SGB2
    }
}

// VALU reads of MFMA results whose producers the compiler did not order for us (the rescale asm writes, then the epilogue reads): 24 wait states
__device__ __forceinline__ void mfma_result_fence(f16v& a, f16v& b) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b));
}

template <int MINB, int ABL, bool KSPLIT = false>
__global__ __launch_bounds__(256, MINB) void flash_attn_x3b_kernel(const AttnParams p) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int NW = 4, QPB = NW * QPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ext_s = reinterpret_cast<float*>(smem + LDS_EXT);               // [257]: bias by clamp(i - 128, +-64), pre-multiplied by log2(e)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 31, hh = lane >> 5;
    const int nqb = (p.T + QPB - 1) / QPB;
    // KSPLIT: the keys of a (head, sample, query block) are cut into p.ksplit ranges of whole 64-key tiles, one workgroup each (adjacent
    // ids: one XCD, one K / V image); launches of <= 2 samples only (batch 1: a wave's 30-block serial chain is what a launch takes)
    const int S = KSPLIT ? p.ksplit : 1;
    const int Lall = xcd_remap(blockIdx.x, gridDim.x);
    const int Lid = KSPLIT ? Lall / S : Lall, zsp = KSPLIT ? Lall - Lid * S : 0;
    const int qb = Lid % nqb, hb = Lid / nqb;
    // ids in (head, sample, query block) order: an XCD's contiguous share of the grid is two heads of EVERY sample.  In (sample, head, ...)
    // order an XCD held ONE sample's 16 heads, so a ragged batch took as long as its longest row (attention work ~ len^2) while the
    // other XCDs idled; the K / V image of a (sample, head) is still shared by adjacent ids (its query blocks).
    const int b = hb % p.B, h = hb / p.B;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = qb * QPB;
    if (q0 >= len) return;
    const int Tq = AttnPlanes::tq(p.T);
    const unsigned char* himg = static_cast<const unsigned char*>(p.planes) + ((size_t)b * p.H + h) * AttnPlanes::head_bytes(p.T);
    const unsigned char* kvimg = himg + AttnPlanes::q_bytes(p.T);
    for (int i = tid; i < EXT_N; i += NW * 64) {
        int o = i - EXT_HALF;
        o = o < -BIAS_CLIP ? -BIAS_CLIP : (o > BIAS_CLIP ? BIAS_CLIP : o);
        ext_s[i] = p.bias_tab[h * (2 * BIAS_CLIP + 1) + o + BIAS_CLIP] * LOG2E;
    }
    if (tid < 16) {      // the four fragment chunks of the ones area: 16.0 (fp16 x 2) at plane 0 of both key steps, zeros at plane 1
        const int c = tid >> 2;       // chunk: (plane, key step)
        reinterpret_cast<unsigned*>(smem + LDS_ONES + (c >> 1) * VHALF + (c & 1) * (2 * D * 16))[tid & 3] = (c >> 1) ? 0u : 0x4C004C00u;
    }
    const float bias_lo = p.bias_tab[h * (2 * BIAS_CLIP + 1)] * LOG2E, bias_hi = p.bias_tab[h * (2 * BIAS_CLIP + 1) + 2 * BIAS_CLIP] * LOG2E;

    const int tq0 = q0 + wave * QPW, t = tq0 + q;
    const int ntiles_all = (len + KT - 1) / KT;
    // this workgroup's tiles [jt0, jt0 + ntiles) of the sample's ntiles_all; everything below counts tiles and blocks from jt0 (`boff` =
    // the absolute index of local block 0 enters where a key's POSITION matters: bias classes, bias table, length mask)
    const int jt0 = KSPLIT ? zsp * ntiles_all / S : 0;
    const int ntiles = KSPLIT ? (zsp + 1) * ntiles_all / S - jt0 : ntiles_all, nblk = 2 * ntiles, boff = 2 * jt0;
    const bool wave_active = tq0 < len;

    // LDS-DMA of what iteration j1 needs (issued one iteration ahead): K tile j1 and V tile j1 - 1 = blocks 2 j1 - 2, 2 j1 - 1 (the PV
    // product runs two blocks behind QK^T): 24 pieces of 1 KiB, six per wave - K pieces wave, wave + 4, wave + 8 and the three pieces
    // of plane (wave & 1) of V block 2 j1 - 2 + (wave >> 1).  K stage = tile parity; a V block lives in ring slot (block & 3) as
    // [plane][j][hh][channel].  Every piece is unconditionally live: a tile index outside the sequence is clamped (the piece lands in
    // a stage / slot nobody reads before it is overwritten).
    const unsigned lane16 = lane * 16;
    const unsigned char* lane_img = kvimg + lane16 + (size_t)jt0 * AttnPlanes::TILE_BYTES;
    auto dma_group = [&](int j1) __attribute__((always_inline)) {
        const int kt = j1 < ntiles ? j1 : ntiles - 1, vt = j1 < 1 ? 0 : j1 - 1;
        const unsigned char* ksrc = lane_img + (size_t)kt * AttnPlanes::TILE_BYTES + wave * 1024;
        const unsigned kdst = LDS_K + (j1 & 1) * KBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + i * 4096),
                                             (__attribute__((address_space(3))) void*)(smem + kdst + i * 4096), 16, 0, 0);
        const int u = wave >> 1, pl = wave & 1, blk = 2 * j1 - 2 + u;
        const unsigned char* vsrc = lane_img + (size_t)vt * AttnPlanes::TILE_BYTES + KBYTES + pl * VPLANE_G + u * VHALF;
        const unsigned vdst = LDS_V + (blk & 3) * VSLOT + pl * VHALF;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc + i * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + vdst + i * 1024), 16, 0, 0);
    };

    WaveState st;
    {
        const int tc = t < Tq ? t : Tq - 1;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                st.qf[s][pl] = as_hf(*reinterpret_cast<const uint4*>(himg + ((size_t)(pl * (D / 8) + 2 * s + hh) * Tq + tc) * 16));
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) st.oacc[ct][r] = 0.f;
    st.m_run = -INFINITY;
    st.alpha = 1.f;
    st.need = 0;

    const bool empty = KSPLIT && ntiles == 0;          // fewer tiles than splits: this range has no keys (workgroup-uniform)
    if (!empty) {
    dma_group(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }

    // per-lane LDS byte offsets: K chunk (plane 0, c8 = hh, key q); V chunks (plane 0, j 0, hh, channel vch) of a ring slot: channel
    // ct 32 + q; lanes q >= 16 of the second channel tile (accumulator rows 48..63) read the ones area, whatever the slot: row 48 = sum of P
    const unsigned k_lane = LDS_K + (hh * KT + q) * 16;
    const int vch0 = q, vch1 = 32 + (q & 15);
    const unsigned v_lane0 = LDS_V + (hh * D + vch0) * 16, v_lane1 = q < 16 ? LDS_V + (hh * D + vch1) * 16 : LDS_ONES;
    const unsigned slot_mask1 = q < 16 ? ~0u : 0u;

    // step b: QK^T of block b + 1, PV of block b - 1, softmax of block b with the FAR or the GENERAL code (gen)
    auto step = [&](int bs, auto par, auto do_qk, auto do_pv, auto do_sm, auto gen) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool DQ = decltype(do_qk)::value, DP = decltype(do_pv)::value, DS = decltype(do_sm)::value, GEN = decltype(gen)::value;
        const int bq = bs + 1, bv = bs - 1;
        const unsigned kb_addr = k_lane + ((bq >> 1) & 1) * KBYTES + (bq & 1) * (32 * 16);
        const unsigned voff = (bv & 3) * VSLOT;
        const unsigned va0 = v_lane0 + voff, va1 = v_lane1 + (voff & slot_mask1);
        const int s0b = (bs + boff) * 32;
        const bool far_hi = s0b - (tq0 + QPW - 1) >= BIAS_CLIP, far_lo = (s0b + 31) - tq0 <= -BIAS_CLIP;
        const int lim = len - s0b - 4 * hh;                                  // key of register r is valid iff roff(r) < lim
        const float bfar = far_hi ? bias_hi : bias_lo;
        const float* ext_lane = ext_s + (s0b + 4 * hh - t + EXT_HALF);       // near blocks only: in range there
        attn_step<PAR, DQ, DP, DS, GEN, ABL>(st, smem, kb_addr, va0, va1, !(far_hi || far_lo), ext_lane, lim, bfar);
    };
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // Waves whose 32 queries all lie beyond the length only stage their share of the tiles (same barriers); the active ones run the
    // steps.  Two separate loops: a per-iteration `if (active)` is a join with the accumulators live on both sides.
    if (!wave_active && empty) return;
    if (!empty) {
    dma_group(1);
    if (!wave_active) {
        for (int j = 1; j <= ((ABL & 2) ? 1 : ntiles); ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (j < ntiles && !(ABL & 1)) dma_group(j + 1);
        }
        return;
    }
    // iteration 0: QK^T of block 0 alone (block 0 goes to s[0]: "step -1" has odd parity), then step 0 (no PV yet)
    step(-1, P1{}, T1{}, T0{}, T0{}, T1{});
    step(0, P0{}, T1{}, T0{}, T1{}, T1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Iterations 1 .. ntiles - 1 = steps (2 j - 1, 2 j), whose softmax blocks are 2 j - 1 and 2 j.  An iteration is FAR when both blocks lie
    // wholly beyond the bias window on one side and inside the length, else GENERAL; along j that is at most: FAR, GENERAL (the band
    // around the wave's queries: ~4 iterations), FAR, GENERAL (a masked tail).  Each run is its own loop over ONE code path - choosing
    // the path per iteration is a join that costs a copy of the accumulators every time.
    auto iteration = [&](int j, auto gen) __attribute__((always_inline)) {
        if (!(ABL & 1)) dma_group(j + 1);          // K(j + 1) into the stage K(j - 1) left, V(j) into the slots V(j - 2) left
        step(2 * j - 1, P1{}, T1{}, T1{}, T1{}, gen);
        step(2 * j, P0{}, T1{}, T1{}, T1{}, gen);
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of group j + 1 have landed (the barrier covers the others')
            __syncthreads();
        }
    };
    {
        // blocks b with 32 b + 31 - tq0 <= -64 are far below, with 32 b - (tq0 + 31) >= 64 far above; blocks >= bmask reach beyond len
        auto rel = [&](int v) { return v > boff ? v - boff : 0; };                                   // absolute -> local block index
        const int b_lo_end = rel((tq0 - BIAS_CLIP - 31 >= 0) ? (tq0 - BIAS_CLIP - 31) / 32 + 1 : 0);      // first block that is not far below
        const int b_hi_beg = rel((tq0 + QPW - 1 + BIAS_CLIP + 31) / 32);                                  // first block that is far above
        const int bmask = rel(len / 32);                                                                  // first block with a key >= len
        auto clampj = [&](int v) { return v < 1 ? 1 : (v > ntiles ? ntiles : v); };
        // iteration j is FAR-below iff 2 j < b_lo_end, FAR-above iff 2 j - 1 >= b_hi_beg; masked iff 2 j >= bmask
        const int jA = clampj(b_lo_end / 2 + ((b_lo_end & 1) ? 1 : 0));            // first j with 2 j >= b_lo_end  (= ceil(b_lo_end / 2))
        const int jB = clampj((b_hi_beg + 2) / 2);                                 // first j with 2 j - 1 >= b_hi_beg
        const int jM = clampj((bmask + 1) / 2);                                    // first j with 2 j >= bmask
        const int e1 = jA < jM ? jA : jM, e2 = (jB > e1 ? jB : e1) < jM ? (jB > e1 ? jB : e1) : jM;
        for (int j = 1; j < e1; ++j) iteration(j, T0{});
        for (int j = e1; j < e2; ++j) iteration(j, T1{});
        for (int j = e2; j < jM; ++j) iteration(j, T0{});
        for (int j = jM; j < ntiles; ++j) iteration(j, T1{});
    }
    if (ABL & 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // V(ntiles - 1) arrived with the last group: the last softmax with the PV product before it, then the PV product of the last block
    step(nblk - 1, P1{}, T0{}, T1{}, T1{}, T1{});
    step(nblk, P0{}, T0{}, T1{}, T0{}, T1{});
    }      // !empty

    mfma_result_fence(st.oacc[0], st.oacc[1]);
    if (KSPLIT && S > 1) {
        // This wave's (O 24 values, l, m) per lane -> its slab (lane-contiguous runs: 26 x 256 floats per workgroup); the LAST wave of
        // the S to arrive merges them in split order, its own included, from memory - the result never depends on who arrives last:
        // O = sum_z O_z 2^(m_z - M), l likewise, M = max_z m_z (m is the lazy running maximum the P of that range were taken against,
        // log2 domain; an empty range left m = -inf, O = l = 0).  Agent-scope (sc1) accesses as in conv_x3's split-K: coherent across the
        // XCDs without a release / acquire fence pair's bulk L2 write-back.  Per WAVE: no workgroup barrier, inactive waves never arrive.
        float* slab = p.kpart + ((size_t)Lid * S) * X3_SLAB_FLOATS;
        float* mine = slab + (size_t)zsp * X3_SLAB_FLOATS + tid;
#pragma unroll
        for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + r * 256, st.oacc[0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int r = 0; r < 9; ++r) __hip_atomic_store(mine + (16 + r) * 256, st.oacc[1][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 25 * 256, st.m_run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have reached the coherent level
        int old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(p.kcount + Lid * NW + wave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != S - 1) return;
        if (lane == 0) __hip_atomic_store(p.kcount + Lid * NW + wave, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
        float M = -INFINITY;
        for (int zz = 0; zz < S; ++zz)
            M = fmaxf(M, __hip_atomic_load(slab + (size_t)zz * X3_SLAB_FLOATS + tid + 25 * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int r = 0; r < 16; ++r) st.oacc[0][r] = st.oacc[1][r] = 0.f;
        for (int zz = 0; zz < S; ++zz) {
            const float* src = slab + (size_t)zz * X3_SLAB_FLOATS + tid;
            const float mz = __hip_atomic_load(src + 25 * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float sc = mz == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mz - M);
#pragma unroll
            for (int r = 0; r < 16; ++r) st.oacc[0][r] = fmaf(sc, __hip_atomic_load(src + r * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), st.oacc[0][r]);
#pragma unroll
            for (int r = 0; r < 9; ++r) st.oacc[1][r] = fmaf(sc, __hip_atomic_load(src + (16 + r) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), st.oacc[1][r]);
        }
    }
    if (t >= len) return;
    const float inv = 1.f / st.oacc[1][8];              // acc = (1024 P)(16 V); row 48 = (1024 P)(16): the denominator at V's scale
    // lane (q, hh) holds channels ct 32 + 8 rg + 4 hh + (0..3) of query t: half hh of the 8-channel chunk ct 4 + rg
    if (p.out_x3) {
        unsigned char* ob = static_cast<unsigned char*>(p.out_x3) + ((long long)b * (p.H * D / 8) * NPL) * p.x3_tp * 16;
        const float sx = inv * XS_SCALE_X;              // the conv's activation planes carry XS_SCALE_X
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int rg = 0; rg < (ct ? 2 : 4); ++rg) {
                unsigned w0[2], w1[2];
                split_pair(st.oacc[ct][4 * rg] * sx, st.oacc[ct][4 * rg + 1] * sx, w0[0], w1[0]);
                split_pair(st.oacc[ct][4 * rg + 2] * sx, st.oacc[ct][4 * rg + 3] * sx, w0[1], w1[1]);
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
            for (int e = 0; e < 4; ++e) ob[(long long)(ct * 32 + 8 * rg + 4 * hh + e) * p.o_cs + t] = st.oacc[ct][4 * rg + e] * inv;
}
}  // namespace

// option "attn_ksplit" / DTTS_ATTN_KSPLIT (process-wide, like ln_reg): the largest split tried, 1 = off, 2 .. 4 (default 4);
// option "attn_ksplit_cus" / DTTS_ATTN_KSPLIT_CUS: a launch is split only while its workgroups x S stay within this count (default 256 =
// the CUs; tests raise it to run the split path at any shape)
static std::atomic<int> g_attn_ksplit_cus{-1};
void set_attn_ksplit_cus(int n) { g_attn_ksplit_cus.store(n < 1 ? 1 : n, std::memory_order_relaxed); }
static int attn_ksplit_cus() {
    int v = g_attn_ksplit_cus.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DTTS_ATTN_KSPLIT_CUS");
        set_attn_ksplit_cus(e ? atoi(e) : 256);
        v = g_attn_ksplit_cus.load(std::memory_order_relaxed);
    }
    return v;
}
static std::atomic<int> g_attn_ksplit{-1};
void set_attn_ksplit(int n) { g_attn_ksplit.store(n < 1 ? 1 : (n > 4 ? 4 : n), std::memory_order_relaxed); }
int attn_ksplit() {
    int v = g_attn_ksplit.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DTTS_ATTN_KSPLIT");
        set_attn_ksplit(e ? atoi(e) : 4);
        v = g_attn_ksplit.load(std::memory_order_relaxed);
    }
    return v;
}

// operands = AttnPlanes images (p.planes)
void launch_flash_attention_x3b(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out && p.planes, "attention_x3b covers head dim 48 with the T5 bias on operand images");
    constexpr int NW = 4;
    // workgroups per CU: 2 (218 registers; 1024 workgroups of the headline launch = two full rounds of 512) or, DTTS_ATTN_OCC=3, 3 (spills)
    static const int occ = []() { const char* v = getenv("DTTS_ATTN_OCC"); return v ? atoi(v) : 2; }();
    const int base = cdiv(p.T, NW * QPW) * p.H * p.B;
    static const int abl = []() { const char* v = getenv("DTTS_ATTN_ABLATE"); return v ? atoi(v) : 0; }();
    // Key split (round 6): a launch of <= 2 samples (the batch-1 CFG pair, single-sample unit calls) whose (head, sample, 128-query)
    // workgroups do not even fill the CUs cuts the keys into S ranges, one workgroup each, merged by the last wave to arrive - but only
    // while base x S workgroups still find a CU each: at T = 936 the pair's 256 workgroups already occupy every CU and a split buys
    // nothing (measured: batch-1 diffusion 122.8 ms without, 123.4 / 125.9 / 126.8 ms with S = 2 / 3 / 4, profiles/r06_batch1.txt), so
    // the headline shapes run exactly round 5's launch.  DTTS_ATTN_KSPLIT = the largest S tried (default 4; 1 = off).
    const int ks_env = attn_ksplit();
    static const int ks_maxb = []() { const char* v = getenv("DTTS_ATTN_KSPLIT_MAXB"); return v ? atoi(v) : 2; }();
    const int ks_cus = attn_ksplit_cus();
    int S = 1;
    if (!abl && p.B <= ks_maxb && ks_env > 1) {
        S = ks_env;
        while (S > 1 && ((long long)base * S > ks_cus || AttnPlanes::nt64(p.T) < 2 * S || (size_t)base * S > X3_MAX_SLABS || (size_t)base * NW > X3_SPLIT_COUNTERS)) --S;
    }
    const dim3 grid(base * S);
    auto go = [&](auto kern) {
        lds_optin(reinterpret_cast<const void*>(kern), LDS_BYTES);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS_BYTES, stream, p);
    };
    if (S > 1) {
        AttnParams q = p;
        q.ksplit = S;
        x3_split_workspace(stream, (size_t)base * S, &q.kpart, &q.kcount);
        auto kern = flash_attn_x3b_kernel<2, 0, true>;
        lds_optin(reinterpret_cast<const void*>(kern), LDS_BYTES);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS_BYTES, stream, q);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (abl) {          // measurement builds (garbage results): which ingredient of the loop costs what (DESIGN.md par. 4)
        switch (abl) {
            case 1: go(flash_attn_x3b_kernel<2, 1>); break;
            case 2: go(flash_attn_x3b_kernel<2, 2>); break;
            case 3: go(flash_attn_x3b_kernel<2, 3>); break;
            case 4: go(flash_attn_x3b_kernel<2, 4>); break;
            case 8: go(flash_attn_x3b_kernel<2, 8>); break;
            case 16: go(flash_attn_x3b_kernel<2, 16>); break;
            case 24: go(flash_attn_x3b_kernel<2, 24>); break;
            case 28: go(flash_attn_x3b_kernel<2, 28>); break;
            case 32: go(flash_attn_x3b_kernel<2, 32>); break;
            case 35: go(flash_attn_x3b_kernel<2, 35>); break;
            case 39: go(flash_attn_x3b_kernel<2, 39>); break;
            default: DTTS_REQUIRE(false, "DTTS_ATTN_ABLATE: 1, 2, 3, 4, 8, 16, 24, 28, 32, 35 or 39");
        }
    } else if (occ == 2) {
        go(flash_attn_x3b_kernel<2, 0>);
    } else {
        go(flash_attn_x3b_kernel<3, 0>);
    }
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
