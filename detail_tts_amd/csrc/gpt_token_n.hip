// hipcc-flags: -fno-slp-vectorize
// (no packed fp32 math, like gpt_token.hip - see "CU sharing" in gpt_token_dev.h)
// The persistent decode-token kernel on a QUARTER or HALF of the workgroups (gpt/model.py:107-185 GPT2InferenceModel.forward with the
// KV cache; same maths as gpt_token.hip, which documents the exchange protocol and the work split).
//
// Why.  Under the request pipeline (SynthesizerTrn.infer_stream) stage A of request i + 1 decodes next to the diffusion trunk of request
// i, and what it costs the trunk is CU-time: gpt_token_kernel holds 128 CUs (314 VGPRs per lane: two of a CU's three conv workgroups
// have to leave) for the whole latency chain of a token, although a workgroup's own work is ~20 us of a 27 us layer and most of THAT is
// per-workgroup overhead (two LayerNorms of all rows, the tile writes, barriers, polls) that every one of the 128 workgroups repeats.
// Here TGN = 128 / NV workgroups (NV = 2 or 4) each run NV of the 128 "virtual workgroups" of gpt_token.hip:
//   * what is the same for every virtual workgroup - the polls of X / AT / Y, both LayerNorms, the LDS activation tile - is done once
//     per REAL workgroup;
//   * the column GEMVs of the NV virtual workgroups are fused: one pass over the tile's k rows feeds NV weight words per thread;
//   * the weights no longer fit in registers a phase ahead (NV x 56 float4 per thread and layer): they STREAM through a window of D
//     float4 per thread - every consumed word issues the load of the word D positions further down the layer's stream
//     (c_attn | c_proj | c_fc | mlp c_proj, then the next layer's c_attn), so a workgroup keeps D x 4 KB in flight while it computes and
//     nothing but D registers x 4 is ever held.  One CU streams 50 - 70 GB/s (tools/ubench/stream_rate.hip): a layer's 28 MB / TGN per
//     workgroup is 7 / 15 us (NV = 2 / 4), the same order as its FMA issue time;
//   * the window is EMPTY across the attention items (c_attn's rounds request nothing beyond their own words; c_proj's and the head of
//     c_fc's are requested under the AT hop), where the registers hold an item's cached keys (one 256-key round at a time) and V rows
//     instead, prefetched while the previous item finishes.
// Measured (one MI355X, profiles/r06_token_wgs.txt): 80 / 105 / 205 ms per 234 tokens alone on 128 / 64 / 32 workgroups; next to a
// diffusion 64 workgroups make the pipelined step 1.4 % faster than 128 (half the CUs for 1.3 x as long), 32 make it slower: what
// SynthesizerTrn.infer_stream asks for is 64 (dtts_gpt_options.token_wgs).
// Every virtual workgroup computes exactly what it computes in gpt_token.hip - same thread -> (column, k) mapping, same order of every
// sum, same exchange words at the same addresses - so logits, latents and sampled codes are BIT-IDENTICAL to the 128-workgroup kernel
// (tests/test_gpu_gpt.py::test_narrow_token_kernels_equal_the_128_workgroup_kernel_bit_for_bit); a session may even change kernels
// between tokens.  The packed weights are the same arrays (virtual workgroups NV w .. NV w + NV - 1 are consecutive slices); only the
// mlp c_proj gets a packed copy of its own (launch_gpt_token_pack 4), because its 12-byte row loads do not fit the float4 stream.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gpt_kernels.h"

namespace dtts {
namespace {

#include "gpt_token_dev.h"

constexpr int NR = 8;                                             // rows of a session (sessions of <= 4 rows have the 1- / 4-row kernels of gpt_token.hip)
constexpr int Q2 = Q_KT / 2, P2 = P_KT / 2, F2 = F_KT / 2, H2 = H_KT / 2;      // float4 per thread and virtual workgroup: 14, 5, 19, 17
constexpr int W2 = GPT_TOKEN_W2_WORDS;                            // mlp c_proj: 24 rows x 3 columns per thread = 18 float4
static_assert(4 * W2 == 3 * NF, "mlp c_proj words");

template <int NV>
struct NG {
    static constexpr int TGN = TG / NV;
    static constexpr int SQ = 0, SP = SQ + NV * Q2, SF = SP + NV * P2, SW = SF + NV * F2, LEN = SW + NV * W2;      // stream positions of a layer
    static constexpr int D = NV == 2 ? 28 : GPT_TOKEN_N_WINDOW;   // weight words in flight per thread (NV = 2: 14 rounds = ~1.5 us of FMAs ahead, the HBM latency under load)
    static constexpr int HLEN = 3 * NV * H2;                      // mel_head: 3 passes
    static_assert(D <= NV * Q2 && D <= NV * H2, "the window reaches into the next layer's c_attn only");
};

template <int NV>
struct SmemN {
    alignas(16) float xs[NR / 4][KP][4];                          // activation tile (P2: the PV partials; P3 / P5: regroup slot B)
    alignas(16) float red[NV * 4096];                             // k-lane partials of NV fused GEMVs | attention scores | regroup slot A
    alignas(16) float qkv[3][TD];
    alignas(16) float hs[NV][NF][NR];
    float own_x[NV][Geo<NR>::RS_PER], own_y[NV][Geo<NR>::RS_PER];
    alignas(16) float st1[NR][4];
    alignas(16) float st2[NR][4];
    float part[NV][4][Geo<NR>::RS_PER];
    float oq[NV][NR * NF];
    float mred[4], lred[4];
};
static_assert(sizeof(float) * 6144 <= sizeof(float4) * KP * (NR / 4) && 6144 <= 2 * 4096, "regroup slots");
static_assert(sizeof(SmemN<4>) <= 112 * 1024, "LDS: one 41 KB conv workgroup still fits beside a token workgroup");

// An opaque zero in SCALAR registers: added to a uniform base pointer the sum stays uniform, so the loads behind it keep the
// `global_load v, v_offset, s[base]` form (one 32-bit lane offset for all of them) instead of a 64-bit VALU add per load.
__device__ __forceinline__ long long opaque_zero() {
    long long z = 0;
    asm volatile("" : "+s"(z)::"memory");
    return z;
}
// The accumulators of a round pass through an empty volatile asm: volatile asms keep their program order, so the FMAs that produce the
// operands cannot sink below it and those of the next round cannot rise above it.  (A scheduling barrier is not enough: instruction
// selection had already put every FMA of a phase behind all of its tile reads and weight loads - which then spill.)
__device__ __forceinline__ void pin8(float (&a)[8][2], int c) {
    asm volatile("" : "+v"(a[0][c]), "+v"(a[1][c]), "+v"(a[2][c]), "+v"(a[3][c]), "+v"(a[4][c]), "+v"(a[5][c]), "+v"(a[6][c]), "+v"(a[7][c]));
}
__device__ __forceinline__ void pin8x3(float (&a)[8][3], int c) {
    asm volatile("" : "+v"(a[0][c]), "+v"(a[1][c]), "+v"(a[2][c]), "+v"(a[3][c]), "+v"(a[4][c]), "+v"(a[5][c]), "+v"(a[6][c]), "+v"(a[7][c]));
}
// the thread index behind an opaque zero: what a phase derives from it (LDS addresses, k lanes, exchange offsets) is recomputed there
// instead of being computed at the top of the layer and kept in registers - or in scratch - across all the other phases
__device__ __forceinline__ int fresh(int tid) {
    int z = 0;
    asm volatile("" : "+v"(z));
    return tid + z;
}
template <class T>
__device__ __forceinline__ const T* at_z(const T* p, long long z) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + z); }

// word `pos` of a layer's weight stream for real workgroup w (positions >= LEN: the next layer's c_attn)
template <int NV>
__device__ __forceinline__ float4 sload(int pos, const GptTokenLayer& L, const float4* wq_next, int w, int tid, long long z) {
    typedef NG<NV> N;
    if (pos >= N::LEN) {
        const int q = pos - N::LEN, i = q / NV, j = q % NV;
        return ldg4(at_z(wq_next, z) + ((size_t)(NV * w + j) * Q2 + i) * 256 + tid);
    } else if (pos < N::SP) {
        const int q = pos - N::SQ, i = q / NV, j = q % NV;
        return ldg4(at_z(L.wq, z) + ((size_t)(NV * w + j) * Q2 + i) * 256 + tid);
    } else if (pos < N::SF) {
        const int q = pos - N::SP, i = q / NV, j = q % NV;
        return ldg4(at_z(L.wp, z) + ((size_t)(NV * w + j) * P2 + i) * 256 + tid);
    } else if (pos < N::SW) {
        const int q = pos - N::SF, i = q / NV, j = q % NV;
        return ldg4(at_z(L.wf, z) + ((size_t)(NV * w + j) * F2 + i) * 256 + tid);
    } else {
        const int q = pos - N::SW, j = q / W2, i = q % W2;        // virtual workgroup after virtual workgroup
        return ldg4(at_z(L.w2p, z) + ((size_t)(NV * w + j) * W2 + i) * 256 + tid);
    }
}

// The fused column GEMVs of NV virtual workgroups on the LDS tile.  Word (i, j) = s[POS + i NV + j] is col_gemv's wr[i] of virtual
// workgroup j; per virtual workgroup the FMAs, their order, the k-lane partials and their sum are col_gemv's -> the same bits.
// `issue(pos, n)` loads the words D positions behind pos .. pos + n - 1.
template <int NV, int PN, int KL, int KT, int POS, int SN, class Issue>
__device__ __forceinline__ void col_gemv_n(float4 (&s)[SN], Issue&& issue, SmemN<NV>& sm, int tid_in, float (&out)[NV]) {
    constexpr int NC = 2 * PN;
    const int tid = fresh(tid_in);
    static_assert(NR * NC <= 256 && KL * NR * NC <= 4096 && KT % 2 == 0 && POS + NV * (KT / 2) <= SN, "col_gemv_n");
    const int q = tid % PN, kl = tid / PN;
    float acc[NV][NR][2];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int b = 0; b < NR; ++b) acc[j][b][0] = acc[j][b][1] = 0.f;
    // One k pair per round: its NV weight words are consumed (and the NV words D positions further down the stream requested), the
    // tile rows of the NEXT round are read under this round's FMAs; pin8 keeps the rounds apart.
    float x[NR], y[NR];
    auto tile_rows = [&](int i, float (&xo)[NR], float (&yo)[NR]) __attribute__((always_inline)) {
        const int k0 = kl + KL * 2 * i, k1 = k0 + KL;
#pragma unroll
        for (int rq = 0; rq < NR / 4; ++rq) {
            const float4 xa = *reinterpret_cast<const float4*>(sm.xs[rq][k0]), ya = *reinterpret_cast<const float4*>(sm.xs[rq][k1]);
            xo[4 * rq] = xa.x; xo[4 * rq + 1] = xa.y; xo[4 * rq + 2] = xa.z; xo[4 * rq + 3] = xa.w;
            yo[4 * rq] = ya.x; yo[4 * rq + 1] = ya.y; yo[4 * rq + 2] = ya.z; yo[4 * rq + 3] = ya.w;
        }
    };
    tile_rows(0, x, y);
#pragma unroll
    for (int i = 0; i < KT / 2; ++i) {
        issue(POS + i * NV, NV);
        float xn[NR], yn[NR];
        if (i + 1 < KT / 2) tile_rows(i + 1, xn, yn);
        // (per accumulator the order is col_gemv's - the x term, then the y term -; across accumulators all x terms come first, so that the
        // two FMAs of one accumulator are 16 NV instructions apart instead of back to back)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 wv = s[POS + i * NV + j];
#pragma unroll
            for (int b = 0; b < NR; ++b) {
                acc[j][b][0] += x[b] * wv.x;
                acc[j][b][1] += x[b] * wv.y;
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            pin8(acc[j], 0);
            pin8(acc[j], 1);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 wv = s[POS + i * NV + j];
#pragma unroll
            for (int b = 0; b < NR; ++b) {
                acc[j][b][0] += y[b] * wv.z;
                acc[j][b][1] += y[b] * wv.w;
            }
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            pin8(acc[j], 0);
            pin8(acc[j], 1);
        }
        if (i + 1 < KT / 2) {
#pragma unroll
            for (int b = 0; b < NR; ++b) {
                x[b] = xn[b];
                y[b] = yn[b];
            }
        }
    }
    if (kl < KL) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float* r = sm.red + j * 4096 + kl * (NR * NC) + q * 2;
#pragma unroll
            for (int b = 0; b < NR; ++b) *reinterpret_cast<float2*>(r + b * NC) = make_float2(acc[j][b][0], acc[j][b][1]);
        }
    }
    __syncthreads();
    // the k-lane sums of the NV virtual workgroups side by side (per output col_gemv's order: lanes i = 0, 4, 8 .. into a0, 1, 5, 9 ..
    // into a1, .., the remainder into a0, then (a0 + a1) + (a2 + a3)): 8 NV LDS reads in flight per round instead of 4
    float a[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j][0] = a[j][1] = a[j][2] = a[j][3] = 0.f;
    if (tid < NR * NC) {
        const float* r = sm.red + tid;
#pragma unroll 2
        for (int i = 0; i + 4 <= KL; i += 4)
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[j][e] += r[j * 4096 + (i + e) * (NR * NC)];
#pragma unroll
        for (int i = KL / 4 * 4; i < KL; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) a[j][0] += r[j * 4096 + i * (NR * NC)];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) out[j] = (a[j][0] + a[j][1]) + (a[j][2] + a[j][3]);
}

#define NSTAMP(k)                                                                                          \
    do {                                                                                                   \
        if (p.trace && tid == 0 && (w == 0 || w == N::TGN / 2 + 3) && ((k) < 10 || !(p.ablate & 8))) p.trace[((w ? 1 : 0) * 16 + l) * 16 + (k)] = wall_clock64(); \
    } while (0)
// DTTS_GPT_TOKEN_ABLATE bit 3 (with DTTS_GPT_TOKEN_TRACE): slots 10 .. 15 carry stamps from inside the first two attention items instead
#define ASTAMP(k)                                                                                          \
    do {                                                                                                   \
        if (p.trace && tid == 0 && (w == 0 || w == N::TGN / 2 + 3) && (p.ablate & 8)) p.trace[((w ? 1 : 0) * 16 + l) * 16 + (k)] = wall_clock64(); \
    } while (0)

template <int NV>
__device__ __forceinline__ void gpt_token_n_body(const GptTokenParams& p) {
    typedef Geo<NR> G;
    typedef NG<NV> N;
    typedef SmemN<NV> Smem;
    constexpr int RS_PER = G::RS_PER, RS_Q = G::RS_Q, D = N::D, LEN = N::LEN;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
    const int tid_k = threadIdx.x, w = blockIdx.x;
    PollState ps{p.err, (p.ablate & 2) != 0, p.poll_nap};      // (ablate bit 0, no weight loads, is not built into this kernel)
    if (*p.err || (p.ablate & 4)) return;
    const unsigned epoch = *p.epoch;
    const GptCtl* ctl = p.ctl;
    const int B = p.B;
    const Xch xc{__builtin_amdgcn_make_buffer_rsrc(p.xch, (short)0, G::XCH_QUADS * 16, 0x00020000)};
    constexpr int XB = G::X_OFF, QB = G::QKV_OFF, AB = G::AT_OFF, YB = G::Y_OFF, RB = G::RS_OFF;
    float* const slot_a = sm.red;                        // the two 6144-float regroup slots of P3 / P5
    float* const slot_b = &sm.xs[0][0][0];

    for (int k = TC + tid_k; k < KP; k += 256) {
#pragma unroll
        for (int q = 0; q < NR / 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) sm.xs[q][k][e] = 0.f;
    }

    float4 s[LEN + D];                                   // the layer's weight stream as the thread sees it; D words are live at any time
    {
        const GptTokenLayer L0 = p.L[0];
        const long long z = opaque_zero();
#pragma unroll
        for (int e = 0; e < D; ++e) s[e] = sload<NV>(e, L0, L0.wq, w, tid_k, z);
    }
    if (tid_k < NV * 2 * NR) {                            // layer 0's input (the sampler's plain rows) enters the same exchange as every other layer's
        const int vv = NV * w + tid_k / (2 * NR), t = tid_k % (2 * NR), b = t >> 1, h = t & 1;
        const float* x = p.x_in + b * TC + NP * vv + 3 * h;
        q_store(xc, XB + b * XQ + 2 * vv + h, b < B ? x[0] : 0.f, b < B ? x[1] : 0.f, b < B ? x[2] : 0.f, epoch << 4);
    }

    for (int l = 0; l < p.NL; ++l) {
        const GptTokenLayer L = p.L[l];
        const float4* wq_next = p.L[l + 1 < p.NL ? l + 1 : l].wq;
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const int tid = tid_k + zero, lane = tid & 63, wave = (tid >> 6) & 3;
        const unsigned tag = (epoch << 4) | (unsigned)l;
        // words [lo, hi) of the stream are requested (one opaque offset pins the group to its place in the schedule)
        auto issue_range = [&](int lo, int hi) __attribute__((always_inline)) {
            if (lo < hi) {
                const long long z = opaque_zero();
#pragma unroll
                for (int e = lo; e < hi; ++e) s[e] = sload<NV>(e, L, wq_next, w, tid, z);
            }
        };
        // the usual policy: consuming words pos .. pos + n - 1 requests the words D positions further down
        auto issue = [&](int pos, int n) __attribute__((always_inline)) { issue_range(pos + D, pos + D + n); };
        // ... except across the attention items, where the window would only hold registers (keys or V rows + a window do not fit a lane's
        // 168): c_attn's rounds request nothing beyond their own words; c_proj's words and the head of c_fc's are requested after the last
        // item, under the AT hop, which puts the usual policy back in step
        auto issue_q = [&](int pos, int n) __attribute__((always_inline)) { issue_range(pos + D, pos + D + n < N::SP ? pos + D + n : N::SP); };
        // per-thread constants of the layer (before anything else is loaded: vmcnt retires in order)
        float c_bq[NV], c_bf[NV], c_bp[NV], c_b2[3];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            c_bq[j] = tid < NR * NQ ? GLOBAL_PTR(float, L.bq)[NQ * (NV * w + j) + tid % NQ] : 0.f;
            c_bf[j] = tid < NR * NF ? GLOBAL_PTR(float, L.bf)[NF * (NV * w + j) + tid % NF] : 0.f;
            c_bp[j] = tid < RS_PER ? GLOBAL_PTR(float, L.bp)[NP * (NV * w + j) + tid % NP] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) c_b2[j] = tid < NV * 2 * NR ? GLOBAL_PTR(float, L.b2)[NP * (NV * w + tid / (2 * NR)) + 3 * (tid & 1) + j] : 0.f;
        // ------------------------------------------------------------------------------------------------ P1: ln_1 + c_attn
        float v[NR][3];
        q_poll8(xc, [&](int b) { return XB + b * XQ + tid; }, tag, v, ps);
        NSTAMP(0);
        if ((tid >> 1) / NV == w) {                            // the 6 NV columns this workgroup owns: residual for P2b
            const int j = (tid >> 1) % NV;
#pragma unroll
            for (int b = 0; b < NR; ++b)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) sm.own_x[j][b * NP + 3 * (tid & 1) + jj] = v[b][jj];
        }
        ln8<NR>(v, L.g1, L.be1, sm, tid);
        rows_to_tile<NR>(v, sm, tid);
        __syncthreads();
        NSTAMP(10);
        {
            float rq[NV];
            col_gemv_n<NV, Q_PN, Q_KL, Q_KT, N::SQ>(s, issue_q, sm, tid, rq);
            NSTAMP(11);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid < NR * NQ) sm.oq[j][tid] = rq[j] + c_bq[j];
        }
        __syncthreads();
        // attention items (head vv / 8, row vv % 8), vv = NV w + it.  Thread (kq = tid / 4, cgp = tid % 4) holds channels [12 cgp, 12 cgp + 12) of
        // the cached keys 4 (kq + 64 u) .. + 3 in registers; item 0's are requested here, before its q can have arrived, item it + 1's
        // when item it's PV sums are done
        float4 kreg[12];
        // (uniform base + a 32-bit lane offset recomputed behind an opaque zero: `global_load v, v_off, s[base]`, and no per-load 64-bit
        // address that the compiler could hoist out of the item loop and keep in registers)
        // One round u of 256 keys at a time (48 registers): round 0 is prefetched, round 1 - keys 256 .. 511, which only a session's last
        // tokens have - is loaded when round 0's scores are done.
        auto load_k = [&](int it, int u) __attribute__((always_inline)) {
            const int vv = NV * w + it, ah = vv >> 3, ab = vv & 7;
            const bool arow = ab < B;
            const int ncach = arow ? ctl->lp[ab] + ctl->step[ab] - 1 : 0;
            const char* kb = reinterpret_cast<const char*>(p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs + (size_t)(ah * TD) * p.cap) + opaque_zero();
            int oz = 0;
            asm volatile("" : "+v"(oz));
            const int to = tid + oz, kq = to >> 2, cgp = to & 3;
            const int s0 = 4 * (kq + 64 * u);
            if (arow && s0 < ncach) {
#pragma unroll
                for (int c = 0; c < 12; ++c) kreg[c] = ldg4(kb + (unsigned)(((12 * cgp + c) * p.cap + s0) * 4));
            } else {
#pragma unroll
                for (int c = 0; c < 12; ++c) kreg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        // its V rows: thread (slot = tid / 4, cg = tid % 4) owns 12 channels of the keys slot + 64 u; requested with the keys
        float4 vr[6][3];
        auto load_v = [&](int it) __attribute__((always_inline)) {
            const int vv = NV * w + it, ah = vv >> 3, ab = vv & 7;
            const bool arow = ab < B;
            const int ncach = arow ? ctl->lp[ab] + ctl->step[ab] - 1 : 0;
            const char* vb = reinterpret_cast<const char*>(p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs + (size_t)TC * p.cap + ah * TD) + opaque_zero();
            int oz = 0;
            asm volatile("" : "+v"(oz));
            const int to = tid + oz, slot = to >> 2, cg = to & 3;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int sk = slot + 64 * u;
                if (arow && sk < ncach) {
                    const char* src = vb + (unsigned)((sk * TC + cg * 12) * 4);
                    vr[u][0] = ldg4(src);
                    vr[u][1] = ldg4(src + 16);
                    vr[u][2] = ldg4(src + 32);
                } else {
                    vr[u][0] = vr[u][1] = vr[u][2] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        if (tid < NV * (NR * NQ / 3)) {                        // 6 triples per row and virtual workgroup
            const int j = tid / (NR * NQ / 3), t = tid % (NR * NQ / 3), b = t / (NQ / 3), t3 = t - b * (NQ / 3);
            const float* o = sm.oq[j] + b * NQ + 3 * t3;
            q_store(xc, QB + b * (3 * XQ) + (NQ / 3) * (NV * w + j) + t3, o[0], o[1], o[2], tag);
        }
        load_k(0, 0);                                          // (behind the stores: every other workgroup's items wait for those words, and the
        load_v(0);                                             //  42 row loads would sit in front of them in the memory pipeline for ~1 us)
        NSTAMP(1);
        // ------------------------------------------------------------------------------------------------ P2: NV attention items
#pragma unroll 1                                               // (rolled: unrolled, the key / V registers of consecutive items overlap and the kernel needs 366 instead of 257 registers)
        for (int it = 0; it < NV; ++it) {
            const int vv = NV * w + it, ah = vv >> 3, ab = vv & 7;
            const bool arow = ab < B;
            const int an = arow ? ctl->lp[ab] + ctl->step[ab] : 1;
            const int ncach = an - 1;
            const float* cb = p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs;
            const float* kp = cb + (size_t)(ah * TD) * p.cap;
            const int kq = tid >> 2, cgp = tid & 3;
            if (tid < 3 * TD / 3) {
                const int which = tid >> 4, i = tid & 15;
                float f[3];
                q_poll1(xc, QB + ab * (3 * XQ) + which * XQ + ah * (TD / 3) + i, tag, f, ps);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int c = 3 * i + j;
                    sm.qkv[which][c] = which == 0 ? f[j] * 0.14433756729740643f : f[j];
                    if (arow && which > 0) {
                        float* cbw = p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs;
                        if (which == 1) cbw[(size_t)(ah * TD + c) * p.cap + ncach] = f[j];
                        else cbw[(size_t)TC * p.cap + (size_t)ncach * TC + ah * TD + c] = f[j];
                    }
                }
            }
            if (it == 0) NSTAMP(2);
            __syncthreads();
            if (it == 0) ASTAMP(10);
            if (it == 1) ASTAMP(14);
            float* sc = sm.red;
            float mx = -INFINITY;
            if (arow) {
                float qv[12];
#pragma unroll
                for (int c4 = 0; c4 < 3; ++c4) {
                    const float4 qq = *reinterpret_cast<const float4*>(&sm.qkv[0][12 * cgp + 4 * c4]);
                    qv[4 * c4] = qq.x; qv[4 * c4 + 1] = qq.y; qv[4 * c4 + 2] = qq.z; qv[4 * c4 + 3] = qq.w;
                }
                auto score_round = [&](int u) __attribute__((always_inline)) {
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 12; ++c) {
                        a.x += qv[c] * kreg[c].x;
                        a.y += qv[c] * kreg[c].y;
                        a.z += qv[c] * kreg[c].z;
                        a.w += qv[c] * kreg[c].w;
                    }
                    a.x += __shfl_xor(a.x, 1); a.y += __shfl_xor(a.y, 1); a.z += __shfl_xor(a.z, 1); a.w += __shfl_xor(a.w, 1);
                    a.x += __shfl_xor(a.x, 2); a.y += __shfl_xor(a.y, 2); a.z += __shfl_xor(a.z, 2); a.w += __shfl_xor(a.w, 2);
                    const int s0 = 4 * (kq + 64 * u);
                    if (cgp == 0 && s0 < ncach) {
                        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (s0 + e < ncach) {
                                sc[s0 + e] = av[e];
                                mx = fmaxf(mx, av[e]);
                            }
                    }
                };
                score_round(0);
                if (ncach > 256) {                             // (workgroup-uniform)
                    load_k(it, 1);
                    score_round(1);
                }
                for (int sk = 512 + tid; sk < ncach; sk += 256) {
                    float d = 0.f;
#pragma unroll 8
                    for (int c = 0; c < TD; ++c) d += sm.qkv[0][c] * kp[(size_t)c * p.cap + sk];
                    sc[sk] = d;
                    mx = fmaxf(mx, d);
                }
                if (wave == 0) {
                    float d = lane < TD ? sm.qkv[0][lane] * sm.qkv[1][lane] : 0.f;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
                    if (lane == 0) sc[ncach] = d;
                    mx = fmaxf(mx, d);
                }
            }
            const int slot = tid >> 2, cg = tid & 3;
            const float* vp = cb + (size_t)TC * p.cap + ah * TD;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            if (lane == 0) sm.mred[wave] = mx;
            __syncthreads();
            if (it == 0) ASTAMP(11);
            mx = fmaxf(fmaxf(sm.mred[0], sm.mred[1]), fmaxf(sm.mred[2], sm.mred[3]));
            float lsum = 0.f;
            if (arow)
                for (int sk = tid; sk < an; sk += 256) {
                    const float pr = expf(sc[sk] - mx);
                    sc[sk] = pr;
                    lsum += pr;
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
            if (lane == 0) sm.lred[wave] = lsum;
            __syncthreads();
            lsum = (sm.lred[0] + sm.lred[1]) + (sm.lred[2] + sm.lred[3]);
            if (it == 0) ASTAMP(12);
            {
                float acc[12];
#pragma unroll
                for (int c = 0; c < 12; ++c) acc[c] = 0.f;
                if (arow) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        const int sk = slot + 64 * u;
                        const float pr = sk < ncach ? sc[sk] : 0.f;
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            acc[e * 4 + 0] += pr * vr[u][e].x;
                            acc[e * 4 + 1] += pr * vr[u][e].y;
                            acc[e * 4 + 2] += pr * vr[u][e].z;
                            acc[e * 4 + 3] += pr * vr[u][e].w;
                        }
                    }
                    for (int sk = slot + 384; sk < ncach; sk += 64) {
                        const float pr = sc[sk];
                        const float* src = vp + (sk * TC + cg * 12);
#pragma unroll
                        for (int c = 0; c < 12; ++c) acc[c] += pr * src[c];
                    }
                }
                float* pv = slot_b + slot * TD + cg * 12;
#pragma unroll
                for (int e = 0; e < 3; ++e) *reinterpret_cast<float4*>(pv + e * 4) = make_float4(acc[e * 4], acc[e * 4 + 1], acc[e * 4 + 2], acc[e * 4 + 3]);
            }
            if (it + 1 < NV) {                                 // (the registers are free now; the next item's keys and V rows arrive under this item's tail)
                load_k(it + 1, 0);
                load_v(it + 1);
            }
            __syncthreads();
            if (tid < TD) {
                float o = 0.f;
                if (arow) {
                    const float* pv = slot_b + tid;
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
                    for (int q = 0; q < 64; q += 4) {
                        o0 += pv[q * TD];
                        o1 += pv[(q + 1) * TD];
                        o2 += pv[(q + 2) * TD];
                        o3 += pv[(q + 3) * TD];
                    }
                    o = ((o0 + o1) + (o2 + o3) + sc[ncach] * sm.qkv[2][tid]) / lsum;
                }
                sm.oq[0][tid] = o;
            }
            __syncthreads();
            if (tid < TD / 3) q_store(xc, AB + ab * XQ + ah * (TD / 3) + tid, sm.oq[0][3 * tid], sm.oq[0][3 * tid + 1], sm.oq[0][3 * tid + 2], tag);
            __syncthreads();                                   // the item's LDS scratch (q / k / v, scores, PV partials, oq) is reused by the next one
            if (it == 0) ASTAMP(13);
            if (it == 1) ASTAMP(15);
        }
        NSTAMP(3);
        issue_range(N::SP, N::SP + D);                         // the window refills under the AT hop
        // ------------------------------------------------------------------------------------------------ P2b: c_proj + residual
        q_poll8(xc, [&](int b) { return AB + b * XQ + tid; }, tag, v, ps);
        NSTAMP(4);
        rows_to_tile<NR>(v, sm, tid);                          // (the last item's barrier freed the tile)
        __syncthreads();
        {
            float rp[NV];
            col_gemv_n<NV, P_PN, P_KL, P_KT, N::SP>(s, issue, sm, tid, rp);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid < RS_PER) sm.own_y[j][tid] = rp[j] + c_bp[j] + sm.own_x[j][tid];
        }
        __syncthreads();
        if (tid < NV * 2 * NR) {
            const int j = tid / (2 * NR), t = tid % (2 * NR);
            const float* y = sm.own_y[j] + (t >> 1) * NP + 3 * (t & 1);
            q_store(xc, YB + (t >> 1) * XQ + 2 * (NV * w + j) + (t & 1), y[0], y[1], y[2], tag);
        }
        NSTAMP(5);
        // ------------------------------------------------------------------------------------------------ P3: ln_2 + c_fc + gelu + mlp c_proj partials
        q_poll8(xc, [&](int b) { return YB + b * XQ + tid; }, tag, v, ps);
        NSTAMP(6);
        ln8<NR>(v, L.g2, L.be2, sm, tid);                      // (its first barrier also orders the GEMV's tile reads before the rewrite)
        rows_to_tile<NR>(v, sm, tid);
        __syncthreads();
        NSTAMP(12);
        {
            float rf[NV];
            col_gemv_n<NV, F_PN, F_KL, F_KT, N::SF>(s, issue, sm, tid, rf);
            NSTAMP(13);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid < NR * NF) sm.hs[j][tid % NF][tid / NF] = gelu_new(rf[j] + c_bf[j]);
        }
        __syncthreads();
        NSTAMP(14);
#pragma unroll
        for (int j = 0; j < NV; ++j) {                         // virtual workgroup j's [8][768] partial of the mlp output -> its 128 owners
            float acc[NR][3];
#pragma unroll
            for (int b = 0; b < NR; ++b) acc[b][0] = acc[b][1] = acc[b][2] = 0.f;
            float w2f[4 * W2];                                  // rows 24 vv .. + 23 of the mlp c_proj, columns 3 tid .. 3 tid + 2: [row][column]
#pragma unroll
            for (int g = 0; g < 3; ++g) {                       // 8 rows = 6 words at a time
                issue(N::SW + j * W2 + 6 * g, 6);
#pragma unroll
                for (int i = 6 * g; i < 6 * g + 6; ++i) {
                    const float4 t = s[N::SW + j * W2 + i];
                    w2f[4 * i] = t.x; w2f[4 * i + 1] = t.y; w2f[4 * i + 2] = t.z; w2f[4 * i + 3] = t.w;
                }
#pragma unroll
                for (int r = 8 * g; r < 8 * g + 8; ++r) {
#pragma unroll
                    for (int rq4 = 0; rq4 < NR / 4; ++rq4) {
                        const float4 ha = *reinterpret_cast<const float4*>(&sm.hs[j][r][4 * rq4]);
                        const float h[4] = {ha.x, ha.y, ha.z, ha.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int b = 4 * rq4 + e;
                            acc[b][0] += h[e] * w2f[3 * r];
                            acc[b][1] += h[e] * w2f[3 * r + 1];
                            acc[b][2] += h[e] * w2f[3 * r + 2];
                        }
                    }
                }
                pin8x3(acc, 0);
                pin8x3(acc, 1);
                pin8x3(acc, 2);
            }
            // through LDS ([column 0..767][row] is the owners' order for a fixed source), so that a wave's store instruction writes 64
            // consecutive 16-byte words; two slots in turn: virtual workgroup j + 1's FMAs run under j's stores
            float* const slot = (j & 1) ? slot_b : slot_a;
            {
                float* st = slot + 3 * NR * tid;
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int rq4 = 0; rq4 < NR / 4; ++rq4)
                        *reinterpret_cast<float4*>(st + NR * m + 4 * rq4) = make_float4(acc[4 * rq4][m], acc[4 * rq4 + 1][m], acc[4 * rq4 + 2][m], acc[4 * rq4 + 3][m]);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int jj = tid + 256 * i, owner = jj / RS_Q;
                const float* r = slot + 3 * jj;
                q_store(xc, RB + (owner * TG + NV * w + j) * RS_Q + (jj - owner * RS_Q), r[0], r[1], r[2], tag);
            }
        }
        NSTAMP(7);
        // ------------------------------------------------------------------------------------------------ P5: owner sums -> next X
#pragma unroll
        for (int j0 = 0; j0 < NV; j0 += 2) {
            float f[2][NR][3];
            q_poll8(xc, [&](int i) { return RB + (NV * w + j0) * TG * RS_Q + tid + 256 * i; }, tag, f[0], ps);
            q_poll8(xc, [&](int i) { return RB + (NV * w + j0 + 1) * TG * RS_Q + tid + 256 * i; }, tag, f[1], ps);
            if (j0 == 0) NSTAMP(8);
            __syncthreads();                                   // both slots are free: the partials have been stored / the previous pair summed
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    slot_a[3 * (tid + 256 * i) + jj] = f[0][i][jj];
                    slot_b[3 * (tid + 256 * i) + jj] = f[1][i][jj];
                }
            __syncthreads();
            for (int og = tid; og < 2 * 4 * RS_PER; og += 256) {      // per owner 4 groups of 32 sources, then the 4 group sums: a fixed order
                const int which = og / (4 * RS_PER), o4 = og % (4 * RS_PER), o = o4 % RS_PER, g = o4 / RS_PER;
                const float* r = (which ? slot_b : slot_a) + (g * 32) * RS_PER + o;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int sr = 0; sr < 32; sr += 4) {
                    a0 += r[sr * RS_PER];
                    a1 += r[(sr + 1) * RS_PER];
                    a2 += r[(sr + 2) * RS_PER];
                    a3 += r[(sr + 3) * RS_PER];
                }
                sm.part[j0 + which][g][o] = (a0 + a1) + (a2 + a3);
            }
        }
        __syncthreads();
        if (tid < NV * 2 * NR) {                               // slot layout [column][row]; own_y is [row][column]
            const int j = tid / (2 * NR), t = tid % (2 * NR), b = t >> 1, h = t & 1;
            float xn[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int c = 3 * h + jj, o = c * NR + b;
                xn[jj] = ((sm.part[j][0][o] + sm.part[j][1][o]) + (sm.part[j][2][o] + sm.part[j][3][o])) + c_b2[jj] + sm.own_y[j][b * NP + c];
            }
            q_store(xc, XB + b * XQ + 2 * (NV * w + j) + h, xn[0], xn[1], xn[2], ((epoch << 4) | (unsigned)(l + 1)));
        }
        NSTAMP(9);
        __syncthreads();                                       // the slots and `part` are free again
        if (tid < KP - TC) *reinterpret_cast<float4*>(sm.xs[0][TC + tid]) = make_float4(0.f, 0.f, 0.f, 0.f);      // slot B covered the tile's zero rows [768, KP) of row quad 0
#pragma unroll
        for (int e = 0; e < D; ++e) s[e] = s[LEN + e];        // the window now holds the head of the next layer's stream
    }
    // ---------------------------------------------------------------------------------------------------- ln_f, final_norm, mel_head
    {
        const int tid = tid_k, l = p.NL;
        float4 hw[N::HLEN];
        auto hload = [&](int pos, long long z) __attribute__((always_inline)) {      // word (pass, i, j)
            const int pass = pos / (NV * H2), q = pos % (NV * H2), i = q / NV, j = q % NV;
            return ldg4(at_z(p.wh, z) + ((size_t)((NV * w + j) * 3 + pass) * H2 + i) * 256 + tid);
        };
        {
            const long long z = opaque_zero();
#pragma unroll
            for (int e = 0; e < D; ++e) hw[e] = hload(e, z);
        }
        auto hissue = [&](int pos, int n) __attribute__((always_inline)) {
            const long long z = opaque_zero();
#pragma unroll
            for (int e = 0; e < n; ++e)
                if (pos + D + e < N::HLEN) hw[pos + D + e] = hload(pos + D + e, z);
        };
        float v[NR][3];
        q_poll8(xc, [&](int b) { return XB + b * XQ + tid; }, (epoch << 4) | (unsigned)p.NL, v, ps);
        NSTAMP(0);
        ln8<NR>(v, p.lnf_g, p.lnf_b, sm, tid);
        ln8<NR>(v, p.fin_g, p.fin_b, sm, tid);
        if (w == 0) {
#pragma unroll
            for (int b = 0; b < NR; ++b)
                if (b < B) {
                    const int step = ctl->step[b];
                    float* col = (ctl->latents && step < ctl->max_steps) ? ctl->latents + (long long)b * ctl->lat_bs + step : nullptr;
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const int c = 3 * tid + m;
                        p.lat[b * TC + c] = v[b][m];
                        if (col) col[(long long)c * ctl->lat_cs] = v[b][m];
                    }
                }
        }
        rows_to_tile<NR>(v, sm, tid);
        __syncthreads();
        auto head_pass = [&](auto pass_c) __attribute__((always_inline)) {
            constexpr int pass = decltype(pass_c)::value;
            float rh[NV];
            col_gemv_n<NV, H_PN, H_KL, H_KT, pass * NV * H2>(hw, hissue, sm, tid, rh);
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (tid < NR * NH) {
                    const int b = tid / NH, c = 3 * NH * (NV * w + j) + NH * pass + tid % NH;
                    if (b < B) p.logits[(size_t)b * p.Vs + c] = rh[j] + p.bh[c];
                }
            __syncthreads();                                   // `red` reads done before the next pass writes it
        };
        head_pass(std::integral_constant<int, 0>());
        head_pass(std::integral_constant<int, 1>());
        head_pass(std::integral_constant<int, 2>());
        NSTAMP(1);
    }
    if (w == 0 && tid_k == 0) *p.epoch = epoch + 1;
}

template <int NV>
__global__ __launch_bounds__(256) void gpt_token_n_kernel(const GptTokenParams p) { gpt_token_n_body<NV>(p); }
// (Measured and dropped: the 64-workgroup kernel capped at 168 registers per lane - amdgpu_waves_per_eu(3, 3), so that a token workgroup
// would take the place of ONE of a CU's three 168-register conv workgroups instead of two.  With the keys one round at a time and the
// window emptied across the attention items the allocator still parked ~50 long-lived values and two window words per layer in scratch;
// scratch reloads retire in order BEHIND the window's loads: 143 ms per 234 tokens alone instead of 105, 461.5 ms per pipelined step
// instead of 452.7.  profiles/r06_token_wgs.txt)

template <int NV>
bool prepare_one() {
    int nb = 0;
    try {
        lds_optin(reinterpret_cast<const void*>(gpt_token_n_kernel<NV>), LDS_EXCLUSIVE);
    } catch (const Error&) {
        (void)hipGetLastError();
        return false;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gpt_token_n_kernel<NV>), 256, LDS_EXCLUSIVE) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return nb >= 1;
}

}  // namespace

bool gpt_token_n_prepare() { return prepare_one<2>() && prepare_one<4>(); }

// p.wgs = 64 / 32: the token on that many workgroups (sessions of up to 8 rows; bit-identical to launch_gpt_token's 128)
void launch_gpt_token_n(const GptTokenParams& p, hipStream_t s) {
    DTTS_REQUIRE(p.wgs == 64 || p.wgs == 32, "persistent decode token: 128, 64 or 32 workgroups");
    DTTS_REQUIRE(p.B >= 1 && p.B <= NR && p.NL >= 1 && p.NL <= GPT_TOKEN_MAX_LAYERS && p.Vs == GPT_TOKEN_VS, "persistent decode token (narrow): shape");
    DTTS_REQUIRE(p.cap <= 6144, "persistent decode token: KV capacity over the LDS score buffer");
    const int nv = GPT_TOKEN_WGS / p.wgs;
    const int lds = p.exclusive_cu ? LDS_EXCLUSIVE : (int)(nv == 2 ? sizeof(SmemN<2>) : sizeof(SmemN<4>));
    if (nv == 2) hipLaunchKernelGGL(gpt_token_n_kernel<2>, dim3(p.wgs), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(gpt_token_n_kernel<4>, dim3(p.wgs), dim3(256), lds, s, p);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
