// hipcc-flags: -fno-slp-vectorize
// (build.py reads the line above: this file is compiled WITHOUT packed fp32 math - see "CU sharing" below.)
// One decode token of the GPT-2 stack as ONE persistent kernel (gpt/model.py:107-185 GPT2InferenceModel.forward with the KV cache;
// HF GPT2Block: ln_1 -> c_attn -> attention -> c_proj -> + -> ln_2 -> c_fc -> gelu_new -> c_proj -> +; then ln_f, final_norm,
// mel_head: gpt/model.py:41, 173).
//
// Why: the launch-per-GEMV decode step (gpt_kernels.hip) is a chain of 53 dependent launches per token.  Alone that costs ~7.5 us per
// launch; under the diffusion trunk of the previous request (SynthesizerTrn.infer_stream) every launch has to win CUs back from
// resident conv workgroups and the chain stretches 4.5 x (tools/pipeline_trace.py: stage A 465 ms per request vs 103 alone), which
// makes stage A the pipeline's critical stage.  Here the chain never leaves the chip: TG = 128 workgroups stay resident for the whole
// token and hand activations to each other through memory with a low-latency "value + tag" protocol instead of kernel boundaries.
//
// Exchange protocol.  Exchanged activations travel as 16-byte words {3 fp32 values (3 consecutive columns of a row), 32-bit tag},
// written and polled with agent-scope (sc1) 16-byte accesses: coherent across the 8 XCDs' L2s, and self-validating - a consumer spins
// on the words it needs until their tag is (epoch, layer); no barrier, no fence, no flag round trip (one store + one successful load
// per hop).  A lane's aligned 16-byte store reaches memory as one piece (the same property RCCL's LL128 protocol rests on).  `epoch` is a device counter
// the kernel bumps once per launch, so stale words of earlier tokens / sessions never match.  Every word of every buffer is written in
// every launch (rows >= B as zeros).  A poll gives up after SPIN_LIMIT tries and raises the session's error flag (dtts_gpt_finish
// fails loudly) instead of hanging the device.
//
// Work split (C = 768, H = 16, D = 48, F = 3072, rows <= 8; workgroup w of 128, 256 threads):
//   P1  X (all rows, 768) -> ln_1 -> c_attn columns [18 w, 18 w + 18)                                       -> QKV
//   P2  attention of (head w / 8, row w % 8): the cached K rows are in registers BEFORE q arrives, the V rows are loaded while the
//       softmax runs; appends k / v to the cache                                                              -> AT
//   P2b AT (all rows, 768) -> c_proj columns [6 w, 6 w + 6) + bias + X                                        -> Y
//   P3  Y -> ln_2 -> c_fc columns [24 w, 24 w + 24) -> gelu_new -> times rows [24 w, 24 w + 24) of the mlp c_proj: a [8][768] PARTIAL
//       of the mlp output, scattered to the 128 column owners (6 columns each)                               -> RS
//   P5  owner: sum of the 128 partials (fixed order) + bias + Y                                               -> X of the next layer
//   end X -> ln_f -> final_norm -> latents; mel_head columns [66 w, 66 w + 66) in 3 passes                    -> logits (plain stores)
// = 5 exchanges per layer.  A column GEMV keeps its weight slice in REGISTERS (one float4 = 2 columns x 2 k's; k is split over the
// workgroup's threads and combined through LDS), loaded from a bind-time repack in exactly the thread order (tok_pack_kernel) and
// issued one phase AHEAD, before the poll of the phase's input, so the weight stream runs under the exchange latency.  One CU
// streams ~50-70 GB/s (tools/ubench/stream_rate.hip), so the 128 workgroups together pull what the token needs (~340 MB of weights +
// the KV rows) at several TB/s.  fp32 FMA throughout; only the order of the sums differs from the launch-per-GEMV path.
#include <cstdio>
#include <cstdlib>

#include "gpt_kernels.h"

namespace dtts {
namespace {

#include "gpt_token_dev.h"

template <int NR>
__global__ __launch_bounds__(256) void gpt_token_kernel(const GptTokenParams p) {
    typedef Geo<NR> G;
    typedef SmemT<NR> Smem;
    constexpr int RS_PER = G::RS_PER, RS_Q = G::RS_Q;
#ifndef DTTS_TOKEN_NO_SETPRIO                              // (diagnostic builds only: tools/diag_token_pk.py)
    if (p.prio) __builtin_amdgcn_s_setprio(3);                        // under the diffusion trunk: this latency chain's waves issue ahead of the resident conv waves
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(smem_raw);
    const int tid_k = threadIdx.x, w = blockIdx.x;
    PollState ps{p.err, (p.ablate & 2) != 0, p.poll_nap};      // (ablate bit 1: every poll takes whatever it finds)
    const bool no_w = (p.ablate & 1) != 0;
    if (*p.err || (p.ablate & 4)) return;                                   // a timed-out session stays dead (no 0.3 s of spinning per token)
    const unsigned epoch = *p.epoch;
    const GptCtl* ctl = p.ctl;
    const int B = p.B;
    const Xch xc{__builtin_amdgcn_make_buffer_rsrc(p.xch, (short)0, G::XCH_QUADS * 16, 0x00020000)};
    constexpr int XB = G::X_OFF, QB = G::QKV_OFF, AB = G::AT_OFF, YB = G::Y_OFF, RB = G::RS_OFF;      // quad offsets of the buffers in the arena

    for (int k = TC + tid_k; k < KP; k += 256) {
#pragma unroll
        for (int q = 0; q < NR / Smem::RV; ++q)
#pragma unroll
            for (int e = 0; e < Smem::RV; ++e) sm.xs[q][k][e] = 0.f;
    }
    // this workgroup's attention work items: head w / 8, rows w % 8 (+ 8 for a 16-row session: two items, one after the other)
    const int ah = w >> 3;

    float4 wq[Q_KT / 2];
    wload<Q_KT>(wq, p.L[0].wq + (size_t)w * (Q_KT / 2) * 256, tid_k, no_w);
    if (tid_k < 2 * NR) {                                 // layer 0's input (the sampler's plain rows) enters the same exchange as every other layer's
        const int b = tid_k >> 1, h = tid_k & 1;
        const float* x = p.x_in + b * TC + NP * w + 3 * h;
        q_store(xc, XB + b * XQ + 2 * w + h, b < B ? x[0] : 0.f, b < B ? x[1] : 0.f, b < B ? x[2] : 0.f, epoch << 4);
    }

    for (int l = 0; l < p.NL; ++l) {
        const GptTokenLayer L = p.L[l];                        // uniform address: scalar loads
        // thread index behind an opaque zero: nothing derived from it is loop-invariant, so the compiler cannot precompute (and keep
        // live across the layer) the address offsets of every load and store of the body
        int zero = 0;
        asm volatile("" : "+v"(zero));
        const int tid = tid_k + zero, lane = tid & 63, wave = (tid >> 6) & 3;
        const unsigned tag = (epoch << 4) | (unsigned)l;
        // small per-thread constants of the layer, loaded BEFORE the bulk prefetches: vmcnt retires in order, so a bias load issued
        // behind a weight prefetch would wait for all of it
        constexpr int OQ = (NR * NQ + 255) / 256, OF = (NR * NF + 255) / 256;       // outputs per thread of the c_attn / c_fc GEMVs
        float c_bq[OQ], c_bf[OF];
#pragma unroll
        for (int j = 0; j < OQ; ++j) c_bq[j] = tid + 256 * j < NR * NQ ? GLOBAL_PTR(float, L.bq)[NQ * w + (tid + 256 * j) % NQ] : 0.f;
#pragma unroll
        for (int j = 0; j < OF; ++j) c_bf[j] = tid + 256 * j < NR * NF ? GLOBAL_PTR(float, L.bf)[NF * w + (tid + 256 * j) % NF] : 0.f;
        const float c_bp = tid < RS_PER ? GLOBAL_PTR(float, L.bp)[NP * w + tid % NP] : 0.f;
        float c_b2[3];                                         // P5's output triples: thread (row tid / 2, columns 3 (tid % 2) .. + 2)
#pragma unroll
        for (int j = 0; j < 3; ++j) c_b2[j] = tid < 2 * NR ? GLOBAL_PTR(float, L.b2)[NP * w + 3 * (tid & 1) + j] : 0.f;
        // ------------------------------------------------------------------------------------------------ P1: ln_1 + c_attn
        float v[NR][3];
        q_poll8(xc, [&](int b) { return XB + b * XQ + tid; }, tag, v, ps);
        STAMP(0);
        if ((tid >> 1) == w) {                               // the 6 columns this workgroup owns: residual for P2b
#pragma unroll
            for (int b = 0; b < NR; ++b)
#pragma unroll
                for (int j = 0; j < 3; ++j) sm.own_x[b * NP + 3 * (tid & 1) + j] = v[b][j];
        }
        ln8(v, L.g1, L.be1, sm, tid);
        rows_to_tile(v, sm, tid);
        __syncthreads();
        STAMP(10);
        float rq[OQ];
        col_gemv<NR, Q_PN, Q_KL, Q_KT>(wq, sm, tid, rq);
        STAMP(11);
#pragma unroll
        for (int j = 0; j < OQ; ++j)
            if (tid + 256 * j < NR * NQ) sm.oq[tid + 256 * j] = rq[j] + c_bq[j];
        __syncthreads();
        if (tid < NR * NQ / 3) {                              // 6 triples per row
            const int b = tid / (NQ / 3), t3 = tid - b * (NQ / 3);
            const float* o = sm.oq + b * NQ + 3 * t3;
            q_store(xc, QB + b * (3 * XQ) + (NQ / 3) * w + t3, o[0], o[1], o[2], tag);
        }
#pragma unroll 1
        for (int it = 0; it < (NR + 7) / 8; ++it) {            // attention work item (head ah, row ab); a 16-row session has two per workgroup
        const int ab = (w & 7) + 8 * it;
        if (NR < 8 && ab >= NR) continue;                      // 4-row sessions: the workgroups of rows 4 .. 7 have none (workgroup-uniform)
        const bool arow = ab < B;
        const int an = arow ? ctl->lp[ab] + ctl->step[ab] : 1;        // keys including the new one
        const int ncach = an - 1;
        // prefetch for P2 / P2b: the cached keys of this (head, row), the c_proj slice.  K is channel-major with the keys contiguous:
        // thread (kq = tid / 4, cgp = tid % 4) holds channels [12 cgp, 12 cgp + 12) of the 4 consecutive keys 4 (kq + 64 u) .. + 3
        const float* cb = p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs;
        const float* kp = pin_v(cb + (size_t)(ah * TD) * p.cap);
        const int kq = tid >> 2, cgp = tid & 3;
        float4 kreg[2][12];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s0 = 4 * (kq + 64 * u);
            if (arow && s0 < ncach) {
#pragma unroll
                for (int c = 0; c < 12; ++c) kreg[u][c] = ldg4(kp + ((12 * cgp + c) * p.cap + s0));
            } else {
#pragma unroll
                for (int c = 0; c < 12; ++c) kreg[u][c] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        STAMP(1);
        // ------------------------------------------------------------------------------------------------ P2: attention
        if (tid < 3 * TD / 3) {                               // q, k, v of this (row, head): 16 triples each
            const int which = tid >> 4, i = tid & 15;
            float f[3];
            q_poll1(xc, QB + ab * (3 * XQ) + which * XQ + ah * (TD / 3) + i, tag, f, ps);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = 3 * i + j;
                sm.qkv[which][c] = which == 0 ? f[j] * 0.14433756729740643f : f[j];      // q / sqrt(48)
                if (arow && which > 0) {                      // KV append at the row's position
                    float* cbw = p.kv + (size_t)l * p.kv_layer + (size_t)ab * p.kv_bs;
                    if (which == 1) cbw[(size_t)(ah * TD + c) * p.cap + ncach] = f[j];
                    else cbw[(size_t)TC * p.cap + (size_t)ncach * TC + ah * TD + c] = f[j];
                }
            }
        }
        STAMP(2);
        __syncthreads();
        float* sc = sm.red;                                   // scores of this row
        float mx = -INFINITY;
        if (arow) {
            float qv[12];
#pragma unroll
            for (int c4 = 0; c4 < 3; ++c4) {
                const float4 qq = *reinterpret_cast<const float4*>(&sm.qkv[0][12 * cgp + 4 * c4]);
                qv[4 * c4] = qq.x; qv[4 * c4 + 1] = qq.y; qv[4 * c4 + 2] = qq.z; qv[4 * c4 + 3] = qq.w;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < 12; ++c) {
                    a.x += qv[c] * kreg[u][c].x;
                    a.y += qv[c] * kreg[u][c].y;
                    a.z += qv[c] * kreg[u][c].z;
                    a.w += qv[c] * kreg[u][c].w;
                }
                // the 4 channel groups sit in 4 neighbouring lanes
                a.x += __shfl_xor(a.x, 1); a.y += __shfl_xor(a.y, 1); a.z += __shfl_xor(a.z, 1); a.w += __shfl_xor(a.w, 1);
                a.x += __shfl_xor(a.x, 2); a.y += __shfl_xor(a.y, 2); a.z += __shfl_xor(a.z, 2); a.w += __shfl_xor(a.w, 2);
                const int s0 = 4 * (kq + 64 * u);
                if (cgp == 0 && s0 < ncach) {                 // keys >= ncach of the last quad: cache slots not written yet
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (s0 + e < ncach) {
                            sc[s0 + e] = av[e];
                            mx = fmaxf(mx, av[e]);
                        }
                }
            }
            for (int s = 512 + tid; s < ncach; s += 256) {     // long sessions: keys beyond the register-resident rounds
                float d = 0.f;
#pragma unroll 8
                for (int c = 0; c < TD; ++c) d += sm.qkv[0][c] * kp[(size_t)c * p.cap + s];
                sc[s] = d;
                mx = fmaxf(mx, d);
            }
            if (wave == 0) {                                  // the key just produced: q . k from LDS
                float d = lane < TD ? sm.qkv[0][lane] * sm.qkv[1][lane] : 0.f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
                if (lane == 0) sc[ncach] = d;
                mx = fmaxf(mx, d);
            }
        }
        // V rows: thread (slot = tid / 4, cg = tid % 4) owns 12 channels of the keys s = slot + 64 u; issued now, used after the softmax
        const int slot = tid >> 2, cg = tid & 3;
        const float* vp = pin_v(cb + (size_t)TC * p.cap + ah * TD);
        float4 vr[6][3];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = slot + 64 * u;
            if (arow && s < ncach) {
                const float* src = vp + (s * TC + cg * 12);
                vr[u][0] = ldg4(src);
                vr[u][1] = ldg4(src + 4);
                vr[u][2] = ldg4(src + 8);
            } else {
                vr[u][0] = vr[u][1] = vr[u][2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) sm.mred[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(sm.mred[0], sm.mred[1]), fmaxf(sm.mred[2], sm.mred[3]));
        float lsum = 0.f;
        if (arow)
            for (int s = tid; s < an; s += 256) {
                const float pr = expf(sc[s] - mx);
                sc[s] = pr;
                lsum += pr;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
        if (lane == 0) sm.lred[wave] = lsum;
        __syncthreads();
        lsum = (sm.lred[0] + sm.lred[1]) + (sm.lred[2] + sm.lred[3]);
        {
            float acc[12];
#pragma unroll
            for (int c = 0; c < 12; ++c) acc[c] = 0.f;
            if (arow) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int s = slot + 64 * u;
                    const float pr = s < ncach ? sc[s] : 0.f;
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        acc[e * 4 + 0] += pr * vr[u][e].x;
                        acc[e * 4 + 1] += pr * vr[u][e].y;
                        acc[e * 4 + 2] += pr * vr[u][e].z;
                        acc[e * 4 + 3] += pr * vr[u][e].w;
                    }
                }
                for (int s = slot + 384; s < ncach; s += 64) {
                    const float pr = sc[s];
                    const float* src = vp + (s * TC + cg * 12);
#pragma unroll
                    for (int c = 0; c < 12; ++c) acc[c] += pr * src[c];
                }
            }
            float* pv = (NR >= 4 ? &sm.xs[0][0][0] : sm.pvbuf) + slot * TD + cg * 12;
#pragma unroll
            for (int e = 0; e < 3; ++e) *reinterpret_cast<float4*>(pv + e * 4) = make_float4(acc[e * 4], acc[e * 4 + 1], acc[e * 4 + 2], acc[e * 4 + 3]);
        }
        __syncthreads();
        if (tid < TD) {
            float o = 0.f;
            if (arow) {
                const float* pv = (NR >= 4 ? &sm.xs[0][0][0] : sm.pvbuf) + tid;
                float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll 4
                for (int q = 0; q < 64; q += 4) {
                    o0 += pv[q * TD];
                    o1 += pv[(q + 1) * TD];
                    o2 += pv[(q + 2) * TD];
                    o3 += pv[(q + 3) * TD];
                }
                o = ((o0 + o1) + (o2 + o3) + sc[ncach] * sm.qkv[2][tid]) / lsum;      // + the key just produced
            }
            sm.oq[tid] = o;
        }
        __syncthreads();
        if (tid < TD / 3) q_store(xc, AB + ab * XQ + ah * (TD / 3) + tid, sm.oq[3 * tid], sm.oq[3 * tid + 1], sm.oq[3 * tid + 2], tag);
        if (NR > 8) __syncthreads();                          // the item's LDS scratch (q / k / v, scores, PV partials, oq) is reused by the next one
        }
        STAMP(3);
        // prefetch for P2b / P3: the c_proj and c_fc slices
        float4 wp[P_KT / 2];
        wload<P_KT>(wp, L.wp + (size_t)w * (P_KT / 2) * 256, tid, no_w);
        float4 wf[F_KT / 2];
        wload<F_KT>(wf, L.wf + (size_t)w * (F_KT / 2) * 256, tid, no_w);
        // ------------------------------------------------------------------------------------------------ P2b: c_proj + residual
        q_poll8(xc, [&](int b) { return AB + b * XQ + tid; }, tag, v, ps);
        STAMP(4);
        __syncthreads();                                       // the PV partials (aliasing the tile) have been consumed
        rows_to_tile(v, sm, tid);
        __syncthreads();
        float rp[1];
        col_gemv<NR, P_PN, P_KL, P_KT>(wp, sm, tid, rp);
        if (tid < RS_PER) sm.own_y[tid] = rp[0] + c_bp + sm.own_x[tid];
        __syncthreads();
        if (tid < 2 * NR) {
            const float* y = sm.own_y + (tid >> 1) * NP + 3 * (tid & 1);
            q_store(xc, YB + (tid >> 1) * XQ + 2 * w + (tid & 1), y[0], y[1], y[2], tag);
        }
        STAMP(5);
        // ------------------------------------------------------------------------------------------------ P3: ln_2 + c_fc + gelu + mlp c_proj partial
        q_poll8(xc, [&](int b) { return YB + b * XQ + tid; }, tag, v, ps);
        STAMP(6);
        ln8(v, L.g2, L.be2, sm, tid);                          // (its first barrier also orders col_gemv's LDS reads before the tile rewrite)
        rows_to_tile(v, sm, tid);
        __syncthreads();
        STAMP(12);
        float rf[OF];
        col_gemv<NR, F_PN, F_KL, F_KT>(wf, sm, tid, rf);
        STAMP(13);
        float w2[NF][3];                                        // rows [24 w, 24 w + 24) of the mlp c_proj, columns 3 tid .. 3 tid + 2 (12-byte loads)
        {
            typedef float f3 __attribute__((ext_vector_type(3)));
            const float* src = pin_v(L.w2 + (size_t)(NF * w) * TC) + 3 * tid;
            if (no_w) {
#pragma unroll
                for (int j = 0; j < NF; ++j) w2[j][0] = w2[j][1] = w2[j][2] = 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const f3 t = *GLOBAL_PTR(f3, src + j * TC);
                    w2[j][0] = t.x; w2[j][1] = t.y; w2[j][2] = t.z;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < OF; ++j) {
            const int o = tid + 256 * j;
            if (o < NR * NF) sm.hs[o % NF][o / NF] = gelu_new(rf[j] + c_bf[j]);
        }
        __syncthreads();
        STAMP(14);
        {
            float acc[NR][3];
#pragma unroll
            for (int b = 0; b < NR; ++b) acc[b][0] = acc[b][1] = acc[b][2] = 0.f;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if constexpr (NR >= 4) {
#pragma unroll
                    for (int rq4 = 0; rq4 < NR / 4; ++rq4) {
                        const float4 ha = *reinterpret_cast<const float4*>(&sm.hs[j][4 * rq4]);
                        const float h[4] = {ha.x, ha.y, ha.z, ha.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int b = 4 * rq4 + e;
                            acc[b][0] += h[e] * w2[j][0];
                            acc[b][1] += h[e] * w2[j][1];
                            acc[b][2] += h[e] * w2[j][2];
                        }
                    }
                } else {
                    const float h = sm.hs[j][0];
                    acc[0][0] += h * w2[j][0];
                    acc[0][1] += h * w2[j][1];
                    acc[0][2] += h * w2[j][2];
                }
            }
            STAMP(15);
            // The partial goes to the owners as [owner][source][column][row] words.  Through LDS first ([column 0..767][row] is exactly
            // that order for a fixed source), so that a wave's store instruction writes 64 consecutive 16-byte words: full 64-byte lines
            // (24 scattered 8-byte stores per thread cost 17 us per layer: every one a partial-line write-through).
            {
                float* st = sm.red + 3 * NR * tid;             // columns 3 tid .. 3 tid + 2, NR rows each
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if constexpr (NR >= 4) {
#pragma unroll
                        for (int rq4 = 0; rq4 < NR / 4; ++rq4)
                            *reinterpret_cast<float4*>(st + NR * m + 4 * rq4) = make_float4(acc[4 * rq4][m], acc[4 * rq4 + 1][m], acc[4 * rq4 + 2][m], acc[4 * rq4 + 3][m]);
                    } else {
                        st[m] = acc[0][m];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int j = tid + 256 * i, owner = j / RS_Q;
                const float* r = sm.red + 3 * j;
                q_store(xc, RB + (owner * TG + w) * RS_Q + (j - owner * RS_Q), r[0], r[1], r[2], tag);
            }
            // prefetch for the next layer's P1 (unconditional: a conditional reload keeps the old slice live across the whole layer)
            wload<Q_KT>(wq, p.L[l + 1 < p.NL ? l + 1 : l].wq + (size_t)w * (Q_KT / 2) * 256, tid, no_w);
        }
        STAMP(7);
        // ------------------------------------------------------------------------------------------------ P5: owner sum -> next X
        {
            float f[NR][3];
            q_poll8(xc, [&](int i) { return RB + w * TG * RS_Q + tid + 256 * i; }, tag, f, ps);
            STAMP(8);
            __syncthreads();                                   // the transposed partials in `red` have been stored
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) sm.red[3 * (tid + 256 * i) + j] = f[i][j];
        }
        __syncthreads();
        for (int og = tid; og < 4 * RS_PER; og += 256) {       // 4 groups of 32 sources, then the 4 group sums: a fixed order
            const int o = og % RS_PER, g = og / RS_PER;
            const float* r = sm.red + (g * 32) * RS_PER + o;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
            for (int s = 0; s < 32; s += 4) {
                a0 += r[s * RS_PER];
                a1 += r[(s + 1) * RS_PER];
                a2 += r[(s + 2) * RS_PER];
                a3 += r[(s + 3) * RS_PER];
            }
            sm.part[g][o] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        if (tid < 2 * NR) {                                    // slot layout [column][row]; own_y is [row][column]
            const int b = tid >> 1, h = tid & 1;
            float xn[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int c = 3 * h + j, o = c * NR + b;
                xn[j] = ((sm.part[0][o] + sm.part[1][o]) + (sm.part[2][o] + sm.part[3][o])) + c_b2[j] + sm.own_y[b * NP + c];
            }
            q_store(xc, XB + b * XQ + 2 * w + h, xn[0], xn[1], xn[2], ((epoch << 4) | (unsigned)(l + 1)));
        }
        STAMP(9);
        __syncthreads();                                       // `red` is free again
    }
    // ---------------------------------------------------------------------------------------------------- ln_f, final_norm, mel_head
    {
        const int tid = tid_k, l = p.NL;
        float4 wh[H_KT / 2];
        wload<H_KT>(wh, p.wh + (size_t)(w * 3) * (H_KT / 2) * 256, tid, no_w);
        float v[NR][3];
        q_poll8(xc, [&](int b) { return XB + b * XQ + tid; }, (epoch << 4) | (unsigned)p.NL, v, ps);
        STAMP(0);
        ln8(v, p.lnf_g, p.lnf_b, sm, tid);
        ln8(v, p.fin_g, p.fin_b, sm, tid);
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
        rows_to_tile(v, sm, tid);
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            constexpr int OH = (NR * NH + 255) / 256;
            float rh[OH];
            col_gemv<NR, H_PN, H_KL, H_KT>(wh, sm, tid, rh);
            if (pass < 2) wload<H_KT>(wh, p.wh + (size_t)(w * 3 + pass + 1) * (H_KT / 2) * 256, tid, no_w);
#pragma unroll
            for (int j = 0; j < OH; ++j) {
                const int o = tid + 256 * j;
                if (o < NR * NH) {
                    const int b = o / NH, c = 3 * NH * w + NH * pass + o % NH;
                    if (b < B) p.logits[(size_t)b * p.Vs + c] = rh[j] + p.bh[c];
                }
            }
            __syncthreads();                                   // `red` reads done before the next pass writes it
        }
        STAMP(1);
    }
    if (w == 0 && tid_k == 0) *p.epoch = epoch + 1;          // every workgroup read it before it produced anything workgroup 0 waited for
}

// bind-time repack of a K-major weight W[K][CoutP] (N valid columns) into the register order of col_gemv: virtual workgroup v owns
// columns [2 PN v, 2 PN (v + 1)); out[(v KT/2 + i) 256 + t] = {W[k0][n], W[k0][n + 1], W[k1][n], W[k1][n + 1]},
// q = t % PN, kl = t / PN, n = 2 PN v + 2 q, k0 = kl + KL 2 i, k1 = k0 + KL; zeros outside
__global__ void tok_pack_kernel(const float* __restrict__ W, int K, int N, int CoutP, int PN, int KL, int KT2, int NV, float4* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)NV * KT2 * 256) return;
    const int t = (int)(e & 255), i = (int)((e >> 8) % KT2), v = (int)((e >> 8) / KT2);
    const int q = t % PN, kl = t / PN;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (kl < KL)
        for (int h = 0; h < 2; ++h) {
            const int k = kl + KL * (2 * i + h);
            for (int c = 0; c < 2; ++c) {
                const int n = 2 * PN * v + 2 * q + c;
                if (k < K && n < N) r[2 * h + c] = W[(long long)k * CoutP + n];
            }
        }
    out[e] = make_float4(r[0], r[1], r[2], r[3]);
}

// the mlp c_proj W2[3072][768] in the thread order of the 64 / 32-workgroup kernels' weight stream (gpt_token_n.hip): virtual workgroup v
// multiplies rows [24 v, 24 v + 24), thread t columns 3 t .. 3 t + 2; out[(v 18 + i) 256 + t] = floats 4 i .. 4 i + 3 of that [row][column] list
__global__ void tok_pack_w2_kernel(const float* __restrict__ W, float4* __restrict__ out) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)TG * GPT_TOKEN_W2_WORDS * 256) return;
    const int t = (int)(e & 255), i = (int)((e >> 8) % GPT_TOKEN_W2_WORDS), v = (int)((e >> 8) / GPT_TOKEN_W2_WORDS);
    float r[4];
    for (int c = 0; c < 4; ++c) {
        const int f = 4 * i + c, row = f / 3, m = f % 3;
        r[c] = W[(long long)(NF * v + row) * TC + 3 * t + m];
    }
    out[e] = make_float4(r[0], r[1], r[2], r[3]);
}

void pack(const float* W, int K, int N, int CoutP, int PN, int KL, int KT, int NV, float4* out, hipStream_t s) {
    const long long n = (long long)NV * (KT / 2) * 256;
    hipLaunchKernelGGL(tok_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, K, N, CoutP, PN, KL, KT / 2, NV, out);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace

bool gpt_token_supported(int C, int H, int F, int NL, int V) { return C == TC && H == TH && F == TF && NL >= 1 && NL <= GPT_TOKEN_MAX_LAYERS && V <= GPT_TOKEN_VS; }

// Bind-time device check + kernel attributes (per device, outside any stream capture): the 128 workgroups must all be resident at the
// same time, one per CU - fewer CUs (partition modes, CU masks, a 64 KiB-LDS part asked for a whole CU) would run every exchange poll
// into SPIN_LIMIT.  false -> the caller keeps the launch-per-GEMV chain.
bool gpt_token_prepare() {
    if (!device_fits(TG, LDS_EXCLUSIVE)) return false;
    int nb1 = 0, nb4 = 0, nb8 = 0, nb16 = 0;
    try {
        lds_optin(reinterpret_cast<const void*>(gpt_token_kernel<1>), LDS_EXCLUSIVE);
        lds_optin(reinterpret_cast<const void*>(gpt_token_kernel<4>), LDS_EXCLUSIVE);
        lds_optin(reinterpret_cast<const void*>(gpt_token_kernel<8>), LDS_EXCLUSIVE);
        lds_optin(reinterpret_cast<const void*>(gpt_token_kernel<16>), LDS_EXCLUSIVE);
    } catch (const Error&) {
        (void)hipGetLastError();
        return false;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, reinterpret_cast<const void*>(gpt_token_kernel<1>), 256, LDS_EXCLUSIVE) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb4, reinterpret_cast<const void*>(gpt_token_kernel<4>), 256, LDS_EXCLUSIVE) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8, reinterpret_cast<const void*>(gpt_token_kernel<8>), 256, LDS_EXCLUSIVE) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb16, reinterpret_cast<const void*>(gpt_token_kernel<16>), 256, LDS_EXCLUSIVE) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return nb1 >= 1 && nb4 >= 1 && nb8 >= 1 && nb16 >= 1 && gpt_token_n_prepare();
}

size_t gpt_token_pack_floats(int which) {
    switch (which) {
        case 0: return (size_t)TG * (Q_KT / 2) * 256 * 4;
        case 1: return (size_t)TG * (P_KT / 2) * 256 * 4;
        case 2: return (size_t)TG * (F_KT / 2) * 256 * 4;
        case 3: return (size_t)TG * 3 * (H_KT / 2) * 256 * 4;
        default: return (size_t)TG * GPT_TOKEN_W2_WORDS * 256 * 4;      // 4: the mlp c_proj for gpt_token_n.hip
    }
}

void launch_gpt_token_pack(int which, const float* W, int N, int CoutP, float* out, hipStream_t s) {
    float4* o = reinterpret_cast<float4*>(out);
    switch (which) {
        case 0: pack(W, TC, N, CoutP, Q_PN, Q_KL, Q_KT, TG, o, s); break;
        case 1: pack(W, TC, N, CoutP, P_PN, P_KL, P_KT, TG, o, s); break;
        case 2: pack(W, TC, N, CoutP, F_PN, F_KL, F_KT, TG, o, s); break;
        case 3: pack(W, TC, N, CoutP, H_PN, H_KL, H_KT, 3 * TG, o, s); break;
        default: {                                                       // 4: W = the mlp c_proj [3072][768]
            DTTS_REQUIRE(N == TC && CoutP == TC, "token kernel: mlp c_proj shape");
            const long long n = (long long)TG * GPT_TOKEN_W2_WORDS * 256;
            hipLaunchKernelGGL(tok_pack_w2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, o);
            DTTS_CHECK_HIP(hipGetLastError());
        }
    }
}

void launch_gpt_token(const GptTokenParams& p, hipStream_t s) {
    DTTS_REQUIRE(p.B >= 1 && p.B <= GPT_TOKEN_ROWS && p.NL >= 1 && p.NL <= GPT_TOKEN_MAX_LAYERS && p.Vs == GPT_TOKEN_VS, "persistent decode token: shape");
    DTTS_REQUIRE(p.cap <= 6144, "persistent decode token: KV capacity over the LDS score buffer");
    const bool r16 = p.B > 8;                                                          // 9 .. 16 rows: the 16-row instantiation
    const bool r1 = p.B == 1 && p.min_rows <= 1;                                       // one row: its own instantiation (option gpt_token_min_rows = 4 / 8: off)
    const bool r4 = !r1 && p.B <= 4 && p.min_rows <= 4;                                // 1 .. 4 rows: the 4-row one (option gpt_token_min_rows = 8: off)
    const bool narrow = !r16 && !r1 && !r4 && p.wgs != 0 && p.wgs != TG;              // 8-row sessions on 64 / 32 workgroups (gpt_token_n.hip; the same bits)
    const int lds_request = p.exclusive_cu ? LDS_EXCLUSIVE : (int)(r16 ? sizeof(SmemT<16>) : r1 ? sizeof(SmemT<1>) : r4 ? sizeof(SmemT<4>) : sizeof(SmemT<8>));   // the attribute was raised by gpt_token_prepare (bind time)
    // DTTS_GPT_TOKEN_TRACE = n: the n-th launch records wall-clock stamps of workgroups 0 and 37 at every exchange and prints them
    static const int trace_at = []() { const char* v = getenv("DTTS_GPT_TOKEN_TRACE"); return v ? atoi(v) : 0; }();
    static int launches = 0;
    static long long* d_trace = nullptr;
    GptTokenParams q = p;
    q.trace = nullptr;
    const bool tracing = trace_at > 0 && ++launches == trace_at;
    if (tracing) {
        if (!d_trace) DTTS_CHECK_HIP(hipMalloc(&d_trace, sizeof(long long) * 2 * 16 * 16));
        DTTS_CHECK_HIP(hipMemsetAsync(d_trace, 0, sizeof(long long) * 2 * 16 * 16, s));
        q.trace = d_trace;
    }
    if (narrow) launch_gpt_token_n(q, s);
    else if (r16) hipLaunchKernelGGL(gpt_token_kernel<16>, dim3(TG), dim3(256), lds_request, s, q);
    else if (r1) hipLaunchKernelGGL(gpt_token_kernel<1>, dim3(TG), dim3(256), lds_request, s, q);
    else if (r4) hipLaunchKernelGGL(gpt_token_kernel<4>, dim3(TG), dim3(256), lds_request, s, q);
    else hipLaunchKernelGGL(gpt_token_kernel<8>, dim3(TG), dim3(256), lds_request, s, q);
    DTTS_CHECK_HIP(hipGetLastError());
    if (tracing) {
        long long h[2 * 16 * 16];
        DTTS_CHECK_HIP(hipMemcpyAsync(h, d_trace, sizeof(h), hipMemcpyDeviceToHost, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));
        static const char* names[16] = {"X", "QKVst", "qkv", "ATst", "AT", "Yst", "Y", "RSst", "RS", "X'st", "p1tile", "p1gemv", "p3tile", "p3gemv", "gelu", "w2fma"};
        static const int order[16] = {0, 10, 11, 1, 2, 3, 4, 5, 6, 12, 13, 14, 15, 7, 8, 9};
        for (int g = 0; g < 2; ++g) {
            const long long t0 = h[(g * 16) * 16];
            fprintf(stderr, "[gpt_token trace] workgroup %d of %d, us since its layer-0 X poll (100 MHz clock)\n", g ? (narrow ? p.wgs / 2 + 3 : 37) : 0, narrow ? p.wgs : TG);
            for (int l = 0; l <= p.NL; ++l) {
                fprintf(stderr, "  l%-2d", l);
                for (int kk = 0; kk < (l < p.NL ? 16 : 2); ++kk) {
                    const int k = l < p.NL ? order[kk] : kk;
                    fprintf(stderr, " %s %.2f |", l < p.NL ? names[k] : (k ? "end" : "X"), (h[(g * 16 + l) * 16 + k] - t0) * 0.01);
                }
                fprintf(stderr, "\n");
            }
        }
    }
}

}  // namespace dtts
