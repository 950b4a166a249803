// hipcc-flags: -fno-honor-nans
// (build.py reads the line above: no NaN canonicalisation in front of every v_max3_f32; infinities keep their meaning.)
// Split-precision flash attention of the diffusion trunk, block-skewed form (head dim 48, T5 relative-position bias; operands = the
// AttnPlanes images the qkv conv wrote; vqvae/utils/diff_util.py:136-215 AttentionBlock / QKVAttentionLegacy,
// vqvae/utils/xtransformers.py:146-186 RelativePositionBias).
//
// Arithmetic, layouts and the lane <-> key mapping are those of attention_x3w.hip (which this kernel replaces on the product path):
//   S^T[key 32, query 32] += K^T[key, c 16] Q[c 16, query]      3 channel steps x 3 split products   (v_mfma_f32_32x32x16_f16)
//   O[c 32, query 32]     += V[c, key 16] P^T[key 16, query]    2 channel tiles x 2 key steps x 3 split products
// a lane (query q = lane & 31, half hh = lane >> 5) holds the 16 keys (r & 3) + 8 (r >> 2) + 4 hh of its query per 32-key block, and
// those registers are the B operand of the PV product.  What changed is the SCHEDULE (312 vector instructions
// against 42 MFMAs per 64-key tile, matrix pipe busy 34 %):
//   * the unit of work is a 32-key BLOCK, skewed by one block inside a wave: step b issues the 9 QK^T MFMAs of block b + 1 and the 12 PV
//     MFMAs of block b in ONE basic block with the softmax of block b between them, so the vector work of a block sits in the shadow
//     of 21 MFMAs of the same wave (the scheduler cannot move code across the old kernel's per-tile branches); only two 16-register
//     score sets are live (the old form held two whole tiles: 64 registers, and copied one onto the other every tile);
//   * a step is instantiated per block class, chosen by a scalar branch: FAR (every (key, query) pair of the wave's block beyond the
//     bias window on one side: one bucket, no table look-up), FAR + tail mask, NEAR (bias from an LDS table extended to +-128 so the
//     look-up is one ds_read at a constant offset from a per-lane base: no clamp, no address arithmetic per score);
//   * the softmax numerators are split with v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (3 instructions per pair instead of 4);
//   * K tiles are double-buffered, V travels in 32-key half tiles through a ring of four 6 KiB slots (step b needs K of block b + 1
//     and V of block b: half a tile of skew), 24 LDS-DMA pieces per 64 keys as before, one barrier per 64 keys.
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

// The PV products accumulate IN PLACE through tied asm operands.  With the builtin the compiler picked the untied form (vdst != srcC)
// on some paths and moved all 32 accumulator registers once per block (16 v_mov_b64 per step: 1 in 6 of the vector instructions of
// the loop); tied, the accumulators live in one register tuple for the whole kernel.  The compiler does not know these are MFMAs:
//   * operands written by VALU just before (the P planes): one wait state in front (s_nop 1 covers two);
//   * results read by VALU (the rescale, the epilogue): 19 wait states, supplied there by hand (mfma_result_fence / the rescale);
//   * the LDS fragments they read are ordinary register operands: the compiler's own s_waitcnt lgkmcnt covers them.
__device__ __forceinline__ void pv_mfma3(f16v& acc, const hf8 (&a)[NPL], const hf8 (&b)[NPL]) {
    asm("s_nop 1\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %1, %3, %0\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\t"
        "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0"
        : "+v"(acc)
        : "v"(a[1]), "v"(a[0]), "v"(b[0]), "v"(b[1]));
}
__device__ __forceinline__ void mfma_result_fence(f16v& a, f16v& b) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b));
}

struct WaveState {
    hf8 qf[3][NPL];        // Q fragments (B operand of QK^T)
    f16v oacc[2];          // O accumulators: channels 0..31 | 32..47, row 48 (oacc[1][8]) = the denominator, 15 unused rows
    f16v s[2];             // score sets: s[b & 1] holds block b
    float m_run;
};

// One step of a wave: QK^T of block bq (into st.s[bq & 1]) and softmax + PV of block bs = bq - 1 (from st.s[bs & 1]).
//   kb_addr: this lane's byte address of K chunk (plane 0, c8 = hh, key q) of the stage that holds block bq's tile (+ its kb half)
//   v_addr0 / v_addr1: byte address of V chunk (plane 0, j 0, hh, channel vch0 / vch1) of block bs's ring slot; lanes q >= 16 pass the
//   ones area as v_addr1
//   mode: the block class of bs (wave-uniform).  Only the first part - exponent arguments and their maximum - depends on it and is
//   branched; everything that touches the accumulators is ONE code path (a join with the accumulators live on both sides makes the
//   compiler keep them in two register sets and move 32 registers per block).
template <int PAR /* bs & 1 */, bool DO_QK, bool DO_SM, int ABL>
__device__ __forceinline__ void attn_step(WaveState& st, const unsigned char* smem, unsigned kb_addr, unsigned v_addr0, unsigned v_addr1, int mode,
                                          const float* ext_lane, int lim, float bfar) {
    f16v& sq = st.s[PAR ^ 1];
    f16v& ss = st.s[PAR];
    hf8 ka[3][NPL], va[2][2][NPL];                           // K fragments of the 3 channel steps; V fragments [key step][channel tile]
    auto vfrag = [&](int j) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            va[j][0][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + v_addr0 + pl * VHALF + j * (2 * D * 16)));
            va[j][1][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + v_addr1 + pl * VHALF + j * (2 * D * 16)));
        }
    };
    // ABL (measurement builds only, DTTS_ATTN_ABLATE; results are garbage): 1 no LDS-DMA in the loop, 2 no barrier / DMA wait in the
    // loop, 4 no softmax vector work, 8 no PV MFMAs, 16 no QK^T MFMAs, 32 no LDS fragment reads
    if (ABL & 32) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ka[s][pl] = st.qf[s][pl];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) va[j][0][pl] = va[j][1][pl] = st.qf[j][pl];
    }
    if (DO_QK && !(ABL & 32)) {
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ka[s][pl] = as_hf(*reinterpret_cast<const uint4*>(smem + kb_addr + pl * (KCH * 16) + s * (2 * KT * 16)));
    }
    if (DO_SM && !(ABL & 32)) vfrag(0);
    if (DO_QK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[r] = 0.f;
        if (!(ABL & 16)) { DTTS_X3B_MFMA(sq, ka[0], st.qf[0]) }
    }
    // Two wave-uniform branch points on the block class, before and after the rescale.  Each arm only READS the scores and DEFINES
    // fresh values (the maximum and, for NEAR / FAR_MASK, the exponent arguments e; then the P planes): a value that one arm changes in
    // place and another leaves alone would be copied on the arm that leaves it alone (16 registers per block on the FAR path).
    f16v e;                                                  // NEAR / FAR_MASK: exponent arguments (log2 domain), masked; FAR: unused
    if (DO_SM && !(ABL & 4)) {
        float mx;
        if (mode == NEAR) {          // bias from the extended table, tail mask
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(ss[r], SU, ext_lane[roff(r)]);
                e[r] = (roff(r) < lim) ? v : -INFINITY;
            }
            mx = fmaxf(fmaxf(e[0], e[1]), e[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, e[r]), e[r + 1]);
            mx = fmaxf(mx, e[15]);
        } else if (mode == FAR_MASK) {
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = (roff(r) < lim) ? fmaf(ss[r], SU, bfar) : -INFINITY;
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
        // Lazy running maximum, decided per query (both lanes of a query see the pair's maximum): P = exp2(e - m + 10) <= 8192 otherwise.
        // The whole conditional rescale is ONE asm statement with its own skip branch: as a C++ branch it made the compiler keep two
        // copies of the 32 accumulator registers (a phi on the rare path) and move all of them on the path NOT taken.  Inside: the
        // partner lane's maximum (v_permlane32_swap), m_new = the pair's maximum where it exceeds m_run + 2^3, alpha = exp2(m_run -
        // m_new) (0 while m_run = -inf), the accumulators (row 48 = the denominator included) times alpha IN PLACE.  The PV MFMAs of
        // the previous step may still be writing the accumulators: 24 wait states in front of the first multiply (a 32 x 32 MFMA
        // result needs 19 before a VALU read).  Rare: the first block, and whenever a query's maximum grows by more than 2^3.
        {
            const unsigned long long need = __builtin_amdgcn_ballot_w64(mx > st.m_run + M_SLACK);
            float ta, tb;
            f16v& o0 = st.oacc[0];
            f16v& o1 = st.oacc[1];
            asm volatile(
                "s_cmp_eq_u64 %[need], 0\n\t"
                "s_cbranch_scc1 .Lx3b_skip_%=\n\t"
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
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                "v_mul_f32 %0, %[ta], %0\n\tv_mul_f32 %1, %[ta], %1\n\tv_mul_f32 %2, %[ta], %2\n\tv_mul_f32 %3, %[ta], %3\n\t"
                "v_mul_f32 %4, %[ta], %4\n\tv_mul_f32 %5, %[ta], %5\n\tv_mul_f32 %6, %[ta], %6\n\tv_mul_f32 %7, %[ta], %7\n\t"
                "v_mul_f32 %8, %[ta], %8\n\tv_mul_f32 %9, %[ta], %9\n\tv_mul_f32 %10, %[ta], %10\n\tv_mul_f32 %11, %[ta], %11\n\t"
                "v_mul_f32 %12, %[ta], %12\n\tv_mul_f32 %13, %[ta], %13\n\tv_mul_f32 %14, %[ta], %14\n\tv_mul_f32 %15, %[ta], %15\n\t"
                "v_mul_f32 %16, %[ta], %16\n\tv_mul_f32 %17, %[ta], %17\n\tv_mul_f32 %18, %[ta], %18\n\tv_mul_f32 %19, %[ta], %19\n\t"
                "v_mul_f32 %20, %[ta], %20\n\tv_mul_f32 %21, %[ta], %21\n\tv_mul_f32 %22, %[ta], %22\n\tv_mul_f32 %23, %[ta], %23\n\t"
                "v_mul_f32 %24, %[ta], %24\n\tv_mul_f32 %25, %[ta], %25\n\tv_mul_f32 %26, %[ta], %26\n\tv_mul_f32 %27, %[ta], %27\n\t"
                "v_mul_f32 %28, %[ta], %28\n\tv_mul_f32 %29, %[ta], %29\n\tv_mul_f32 %30, %[ta], %30\n\tv_mul_f32 %31, %[ta], %31\n\t"
                "s_nop 3\n"
                ".Lx3b_skip_%=:"
                : "+v"(o0[0]), "+v"(o0[1]), "+v"(o0[2]), "+v"(o0[3]), "+v"(o0[4]), "+v"(o0[5]), "+v"(o0[6]), "+v"(o0[7]), "+v"(o0[8]), "+v"(o0[9]),
                  "+v"(o0[10]), "+v"(o0[11]), "+v"(o0[12]), "+v"(o0[13]), "+v"(o0[14]), "+v"(o0[15]), "+v"(o1[0]), "+v"(o1[1]), "+v"(o1[2]),
                  "+v"(o1[3]), "+v"(o1[4]), "+v"(o1[5]), "+v"(o1[6]), "+v"(o1[7]), "+v"(o1[8]), "+v"(o1[9]), "+v"(o1[10]), "+v"(o1[11]),
                  "+v"(o1[12]), "+v"(o1[13]), "+v"(o1[14]), "+v"(o1[15]), [m] "+v"(st.m_run), [ta] "=&v"(ta), [tb] "=&v"(tb)
                : [mx] "v"(mx), [need] "s"(need)
                : "vcc", "scc");
        }
    }
    if (DO_QK && !(ABL & 16)) {
        DTTS_X3B_MFMA(sq, ka[1], st.qf[1])
        DTTS_X3B_MFMA(sq, ka[2], st.qf[2])
    }
    if (DO_SM) {
        const float m_sub = ((st.m_run == -INFINITY) ? 0.f : st.m_run) - P_SHIFT;
        hf8 pf[2][NPL];                                      // P planes of the two key steps (key step j: this lane's registers 8 j .. 8 j + 7)
        auto split_step = [&](int j, const float (&pv)[8]) {
            uint4 w0, w1;
            split_pair_mix(pv[0], pv[1], w0.x, w1.x);
            split_pair_mix(pv[2], pv[3], w0.y, w1.y);
            split_pair_mix(pv[4], pv[5], w0.z, w1.z);
            split_pair_mix(pv[6], pv[7], w0.w, w1.w);
            pf[j][0] = as_hf(w0);
            pf[j][1] = as_hf(w1);
        };
        if (ABL & 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) pf[j][pl] = st.qf[j][pl];
        } else if (mode == FAR) {
            const float c0 = bfar - m_sub;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = __builtin_amdgcn_exp2f(fmaf(ss[8 * j + k], SU, c0));      // 1024 P: the scale is free in the exponent
                split_step(j, pv);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) pv[k] = __builtin_amdgcn_exp2f(e[8 * j + k] - m_sub);
                split_step(j, pv);
            }
        }
        if (!(ABL & 32)) vfrag(1);
        if (!(ABL & 8)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pv_mfma3(st.oacc[0], va[j][0], pf[j]);
                pv_mfma3(st.oacc[1], va[j][1], pf[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) asm volatile("" ::"v"(pf[j][pl]), "v"(va[j][0][pl]), "v"(va[j][1][pl]));
        }
    }
}

template <int MINB, int ABL>
__global__ __launch_bounds__(256, MINB) void flash_attn_x3b_kernel(const AttnParams p) {
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int NW = 4, QPB = NW * QPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* ext_s = reinterpret_cast<float*>(smem + LDS_EXT);               // [257]: bias by clamp(i - 128, +-64), pre-multiplied by log2(e)

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
    const int ntiles = (len + KT - 1) / KT, nblk = 2 * ntiles;
    const bool wave_active = tq0 < len;

    // LDS-DMA of the data step group j1 needs (issued one group ahead): K tile j1, V blocks 2 j1 - 1 and 2 j1: 24 pieces of 1 KiB, six
    // per wave - K pieces wave, wave + 4, wave + 8 and the three pieces of plane (wave & 1) of V block 2 j1 - 1 + (wave >> 1).  Every
    // piece is unconditionally live: a tile / block index beyond the sequence is clamped to the last one (the piece lands in a stage or
    // slot nobody reads before it is overwritten: the stages and slots of groups j1 - 2 / blocks b - 4).
    const unsigned lane16 = lane * 16;
    const unsigned char* lane_img = kvimg + lane16;
    auto dma_group = [&](int j1) {
        const int kt = j1 < ntiles ? j1 : ntiles - 1;
        const unsigned char* ksrc = lane_img + (size_t)kt * AttnPlanes::TILE_BYTES + wave * 1024;
        const unsigned kdst = LDS_K + (j1 & 1) * KBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc + i * 4096),
                                             (__attribute__((address_space(3))) void*)(smem + kdst + i * 4096), 16, 0, 0);
        const int blk = 2 * j1 - 1 + (wave >> 1), pl = wave & 1;
        const int bc = blk < 0 ? 0 : (blk < nblk ? blk : nblk - 1);
        const unsigned char* vsrc = lane_img + (size_t)(bc >> 1) * AttnPlanes::TILE_BYTES + KBYTES + pl * VPLANE_G + (bc & 1) * VHALF;
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

    dma_group(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane LDS byte offsets: K chunk (plane 0, c8 = hh, key q); V chunks (plane 0, j 0, hh, channel vch): channel ct 32 + q, the
    // padding rows 48..63 of channel tile 1 re-read valid chunks (their accumulator rows are never stored)
    const unsigned k_lane = LDS_K + (hh * KT + q) * 16;
    const int vch0 = q, vch1 = 32 + (q & 15);
    // lanes q >= 16 of the second channel tile (accumulator rows 48..63) read the ones area, whatever the slot: row 48 = sum of P
    const unsigned v_lane0 = LDS_V + (hh * D + vch0) * 16, v_lane1 = q < 16 ? LDS_V + (hh * D + vch1) * 16 : LDS_ONES;
    const unsigned slot_mask1 = q < 16 ? ~0u : 0u;

    // step b: QK^T of block b + 1, softmax + PV of block b; the block class of b (wave-uniform) goes in as `mode`
    auto step = [&](int bs, auto par, auto do_qk, auto do_sm) {
        constexpr int PAR = decltype(par)::value;
        constexpr bool DQ = decltype(do_qk)::value, DS = decltype(do_sm)::value;
        const int bq = bs + 1;
        const unsigned kb_addr = k_lane + ((bq >> 1) & 1) * KBYTES + (bq & 1) * (32 * 16);
        const unsigned va0 = v_lane0 + (bs & 3) * VSLOT, va1 = v_lane1 + (((bs & 3) * VSLOT) & slot_mask1);
        const int s0b = bs * 32;
        const bool far_hi = s0b - (tq0 + QPW - 1) >= BIAS_CLIP, far_lo = (s0b + 31) - tq0 <= -BIAS_CLIP;
        const int mode = !(far_hi || far_lo) ? NEAR : (s0b + 32 > len ? FAR_MASK : FAR);
        const int lim = len - s0b - 4 * hh;                                  // key of register r is valid iff roff(r) < lim
        const float bfar = far_hi ? bias_hi : bias_lo;
        const float* ext_lane = ext_s + (s0b + 4 * hh - t + EXT_HALF);       // NEAR only: in range there
        attn_step<PAR, DQ, DS, ABL>(st, smem, kb_addr, va0, va1, mode, ext_lane, lim, bfar);
    };
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // Waves whose 32 queries all lie beyond the length only stage their share of the tiles (same barriers); the active ones run the
    // steps.  Two separate loops: a per-iteration `if (active)` is a join with the accumulators live on both sides (see attn_step).
    dma_group(1);
    if (!wave_active) {
        for (int j = 1; j <= ((ABL & 2) ? 1 : ntiles); ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (j < ntiles && !(ABL & 1)) dma_group(j + 1);
        }
        return;
    }
    // group 0: QK^T of block 0 alone (block 0 goes to s[0]: "step -1" has odd parity), then step 0
    step(-1, P1{}, T1{}, T0{});
    step(0, P0{}, T1{}, T1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 1; j < ntiles; ++j) {
        if (!(ABL & 1)) dma_group(j + 1);          // K(j + 1) into the stage K(j - 1) left, V(2 j + 1), V(2 j + 2) into the slots V(2 j - 3), V(2 j - 2) left
        step(2 * j - 1, P1{}, T1{}, T1{});
        step(2 * j, P0{}, T1{}, T1{});
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of group j + 1 have landed (the barrier covers the others')
            __syncthreads();
        }
    }
    step(nblk - 1, P1{}, T0{}, T1{});

    mfma_result_fence(st.oacc[0], st.oacc[1]);          // the last PV products are asm: their results are about to be read by VALU
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

// operands = AttnPlanes images (p.planes)
void launch_flash_attention_x3b(const AttnParams& p, hipStream_t stream) {
    DTTS_REQUIRE(p.D == 48 && p.bias_tab && !p.causal && !p.band && !p.ml_out && p.planes, "attention_x3b covers head dim 48 with the T5 bias on operand images");
    constexpr int NW = 4;
    // workgroups per CU: 2 (<= 256 registers; 1024 workgroups of the headline launch = exactly two rounds of 512) or, DTTS_ATTN_OCC=3, 3
    static const int occ = []() { const char* v = getenv("DTTS_ATTN_OCC"); return v ? atoi(v) : 3; }();
    const dim3 grid(cdiv(p.T, NW * QPW) * p.H * p.B);
    static const int abl = []() { const char* v = getenv("DTTS_ATTN_ABLATE"); return v ? atoi(v) : 0; }();
    auto go = [&](auto kern) {
        lds_optin(reinterpret_cast<const void*>(kern), LDS_BYTES);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS_BYTES, stream, p);
    };
    if (abl) {          // measurement builds (garbage results): which ingredient of the loop costs what (DESIGN.md par. 4)
        switch (abl) {
            case 1: go(flash_attn_x3b_kernel<3, 1>); break;
            case 2: go(flash_attn_x3b_kernel<3, 2>); break;
            case 3: go(flash_attn_x3b_kernel<3, 3>); break;
            case 4: go(flash_attn_x3b_kernel<3, 4>); break;
            case 8: go(flash_attn_x3b_kernel<3, 8>); break;
            case 16: go(flash_attn_x3b_kernel<3, 16>); break;
            case 24: go(flash_attn_x3b_kernel<3, 24>); break;
            case 28: go(flash_attn_x3b_kernel<3, 28>); break;
            case 32: go(flash_attn_x3b_kernel<3, 32>); break;
            case 35: go(flash_attn_x3b_kernel<3, 35>); break;
            case 39: go(flash_attn_x3b_kernel<3, 39>); break;
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
