// Device helpers shared by the persistent GPT decode-token kernels: gpt_token.hip (128 workgroups, one virtual workgroup each) and
// gpt_token_n.hip (64 / 32 workgroups that each run 2 / 4 of the same virtual workgroups).  Included inside
// `namespace dtts { namespace {` of those two files only (everything here has internal linkage on purpose).
#pragma once

typedef unsigned long long u64;
// pointers that come out of memory (the layer table) are generic to the compiler: loads through them would be FLAT
#define GLOBAL_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))
typedef float f4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const void* p) {
    const f4e t = *GLOBAL_PTR(f4e, p);
    return make_float4(t.x, t.y, t.z, t.w);
}

constexpr int TG = GPT_TOKEN_WGS, TC = 768, TH = 16, TD = 48, TF = 3072;
constexpr int KP = 864;                       // rows of the LDS activation tile (>= kl + KL * (KT - 1) of every phase; [768, KP) stay 0)
constexpr int SPIN_LIMIT = 1 << 18;
static_assert(TG == 128, "work split");

// column GEMV shapes: PN column pairs per workgroup, KL k-lanes (PN * KL <= 256), KT k's per thread (even; KL * KT >= K)
constexpr int Q_PN = 9, Q_KL = 28, Q_KT = 28;        // c_attn   768 -> 2304 : 18 columns per workgroup
constexpr int P_PN = 3, P_KL = 85, P_KT = 10;        // c_proj   768 ->  768 : 6
constexpr int F_PN = 12, F_KL = 21, F_KT = 38;       // c_fc     768 -> 3072 : 24
constexpr int H_PN = 11, H_KL = 23, H_KT = 34;       // mel_head 768 -> 8448 : 3 passes of 22
constexpr int NQ = 2 * Q_PN, NP = 2 * P_PN, NF = 2 * F_PN, NH = 2 * H_PN;
static_assert(NQ * TG == 3 * TC && NP * TG == TC && NF * TG == TF && 3 * NH * TG == GPT_TOKEN_VS, "column split");
static_assert(Q_KL * Q_KT >= TC && P_KL * P_KT >= TC && F_KL * F_KT >= TC && H_KL * H_KT >= TC, "k split");
static_assert(255 / P_PN + P_KL * (P_KT - 1) < KP && 255 / Q_PN + Q_KL * (Q_KT - 1) < KP && 255 / F_PN + F_KL * (F_KT - 1) < KP &&
                  255 / H_PN + H_KL * (H_KT - 1) < KP, "LDS activation tile");

// exchange arena, in 16-byte words ("quads": 3 consecutive columns of one row + tag)
// Everything below is written for NR rows per session, NR = 8 (round 3), 16 (round 4: two requests of 8 utterances decoded as ONE
// session - the weights stream once per token for both), 4 (round 5: sessions of <= 4 rows - the padded rows of an 8-row launch cost
// their share of every FMA loop, LayerNorm, regroup-and-store tail and exchange word) or 1 (round 5: the batch-1 latency case itself;
// the LDS tile then holds scalars instead of row quads).  Per row
// the arithmetic and the order of every sum are the same in all instantiations, so a row's latents do not depend on which one
// produced them.
constexpr int XQ = TC / 3;                     // quads per row of a 768-wide buffer
template <int NR>
struct Geo {
    static constexpr int X_OFF = 0, QKV_OFF = X_OFF + NR * XQ, AT_OFF = QKV_OFF + NR * 3 * XQ, Y_OFF = AT_OFF + NR * XQ, RS_OFF = Y_OFF + NR * XQ;
    static constexpr int RS_PER = NR * NP;        // values one source sends one owner: 6 columns x NR rows, [column][row]
    static constexpr int RS_Q = RS_PER / 3;       // 16 / 32 quads
    static constexpr int XCH_QUADS = RS_OFF + TG * TG * RS_Q;
    static constexpr int RED = NR <= 8 ? 6144 : 12288;      // floats of the `red` scratch (>= 768 NR; >= the KV capacity: attention scores)
};
static_assert(2 * Geo<16>::XCH_QUADS == GPT_TOKEN_XCH_WORDS && Geo<8>::XCH_QUADS < Geo<16>::XCH_QUADS && Geo<4>::XCH_QUADS < Geo<8>::XCH_QUADS && Geo<1>::XCH_QUADS < Geo<4>::XCH_QUADS, "exchange arena size");

template <int NR>
struct SmemT {
    static constexpr int RV = NR >= 4 ? 4 : NR;                    // rows per tile element (a float4 of 4 rows; NR = 1: a scalar)
    alignas(16) float xs[NR / RV][KP][RV];                         // activation tile [row quad][k][row]: rows 4 q .. 4 q + 3 of input k   (NR >= 4, P2: the PV partials)
    alignas(16) float pvbuf[NR >= 4 ? 4 : 64 * TD];                // NR = 1: the PV partials' own buffer (the scalar tile is too small to alias)
    alignas(16) float red[Geo<NR>::RED];     // k-lane partials of a column GEMV | gathered mlp partials | attention scores
    alignas(16) float qkv[3][TD];
    alignas(16) float hs[NF][NR];   // gelu(c_fc) of this workgroup's 24 columns, [column][row]
    float own_x[Geo<NR>::RS_PER], own_y[Geo<NR>::RS_PER];      // residual rows of the 6 columns this workgroup owns
    alignas(16) float st1[NR][4];  // LayerNorm: per-row wave partials
    alignas(16) float st2[NR][4];
    float part[4][Geo<NR>::RS_PER];
    float oq[NR * NF < 64 ? 64 : NR * NF];      // a phase's outputs, regrouped into triples before they are stored (>= 48: the attention's output row)
    float mred[4], lred[4];
};
static_assert(sizeof(float) * 64 * TD <= sizeof(float4) * KP, "PV partials alias the activation tile (NR >= 4)");
// CU sharing.  Built with packed fp32 math (`v_pk_fma_f32`, the SLP vectoriser's default), token workgroups that shared a CU with the
// diffusion trunk's split-precision conv / attention workgroups (stage B of the previous request under SynthesizerTrn.infer_stream)
// gave WRONG results - deterministic alone, a few accumulators of some workgroups off by percents under that load, sampled codes
// changed; not under a rocBLAS load, not under the exact-fp32 conv kernels.  An LDS canary next to the same load saw no foreign
// write; FLAT / inline-asm accesses and AGPR use were ruled out; the first corrupted values traced to the hi lane of
// `v_pk_fma_f32 ... op_sel_hi:[1,0,1]` in col_gemv.  Round 3 shipped packed math + an occupancy trick (request the CU's whole LDS so
// that nothing with LDS co-resides); zero-LDS kernels could still share the CU.  Since round 4 the file is built with
// -fno-slp-vectorize (first line; no packed fp32 instruction is left in the object: tests/test_host_logic.py disassembles it), which
// is the variant that was bit-identical in every shared-CU run, AND the exclusive-CU request stays available as a policy knob
// (option "gpt_token_exclusive_cu" / DTTS_GPT_TOKEN_EXCLUSIVE_CU, see DESIGN.md for the measured choice); the stress tests run both
// settings next to LDS kernels, zero-LDS kernels and the vocoder (tests/test_gpu_e2e.py::test_token_kernel_under_concurrent_*).
constexpr int LDS_EXCLUSIVE = 160 * 1024;
static_assert(sizeof(SmemT<1>) <= 64 * 1024 && sizeof(SmemT<4>) <= 64 * 1024 && sizeof(SmemT<8>) <= 64 * 1024 && sizeof(SmemT<16>) <= 128 * 1024, "LDS");

#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        if (p.trace && tid == 0 && (w == 0 || w == 37)) p.trace[((w ? 1 : 0) * 16 + l) * 16 + (k)] = wall_clock64(); \
    } while (0)

// The 16-byte agent-scope accesses are raw buffer loads / stores with the sc1 cache-policy bit: compiler-generated (it tracks their
// latency and their register hazards - hand-written `global_*_dwordx4 ... sc1` inline asm, which it does not, delivered stale store
// data here), one descriptor over the whole arena, 32-bit byte offsets.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;                    // gfx940+ cache policy: bit 0 = sc0, bit 1 = nt, bit 4 = sc1
struct Xch {
    __amdgpu_buffer_rsrc_t rs;
};
__device__ __forceinline__ void q_store(const Xch& x, int q, float a, float b, float c, unsigned tag) {
    const u4v v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), tag};
    __builtin_amdgcn_raw_buffer_store_b128(v, x.rs, q * 16, 0, AUX_SC1);
}

struct PollState {
    int* err;
    bool dead;
    int nap;              // extra s_sleep rounds between two polls (0: poll as fast as possible; stage A has slack under the pipeline)
};
__device__ __forceinline__ void poll_nap(const PollState& ps) {
    __builtin_amdgcn_s_sleep(1);
    for (int i = 0; i < ps.nap; ++i) __builtin_amdgcn_s_sleep(8);
}

// 8 quads idx(i): all loads in flight at once; while any of them is stale, all are read again
template <int N, class F>
__device__ __forceinline__ void q_poll8(const Xch& x, F idx, unsigned tag, float (&out)[N][3], PollState& ps) {
    u4v v[N];
    int spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(x.rs, idx(i) * 16, 0, AUX_SC1);
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) bad |= v[i].w ^ tag;
        if (bad == 0 || ps.dead) break;
        if (++spins > SPIN_LIMIT) {
            ps.dead = true;
            *ps.err = 1;
            break;
        }
        poll_nap(ps);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i][0] = __uint_as_float(v[i].x);
        out[i][1] = __uint_as_float(v[i].y);
        out[i][2] = __uint_as_float(v[i].z);
    }
}
__device__ __forceinline__ void q_poll1(const Xch& x, int q, unsigned tag, float (&out)[3], PollState& ps) {
    u4v v;
    int spins = 0;
    for (;;) {
        v = __builtin_amdgcn_raw_buffer_load_b128(x.rs, q * 16, 0, AUX_SC1);
        if (v.w == tag || ps.dead) break;
        if (++spins > SPIN_LIMIT) {
            ps.dead = true;
            *ps.err = 1;
            break;
        }
        poll_nap(ps);
    }
    out[0] = __uint_as_float(v.x);
    out[1] = __uint_as_float(v.y);
    out[2] = __uint_as_float(v.z);
}

// sums of 8 values per lane over the wave in 10 shuffles (halving exchange): every lane gets the total of row (lane >> 3) & 7
__device__ __forceinline__ float wsum8(const float (&s)[8], int lane) {
    const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
    float t[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (h32 ? s[4 + i] : s[i]) + __shfl_xor(h32 ? s[i] : s[4 + i], 32);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (h16 ? t[2 + i] : t[i]) + __shfl_xor(h16 ? t[i] : t[2 + i], 16);
    float w = (h8 ? u[1] : u[0]) + __shfl_xor(h8 ? u[0] : u[1], 8);
    w += __shfl_xor(w, 4);
    w += __shfl_xor(w, 2);
    w += __shfl_xor(w, 1);
    return w;
}

// the 4-row form of wsum8 (7 shuffles): every lane gets the total of row (lane >> 4) & 3.  Per row the same tree as wsum8 - levels in the
// order 32, 16, 8, 4, 2, 1, every level adds the same two partial sums, and a + b == b + a bit for bit - so a row's statistics do not
// depend on the instantiation
__device__ __forceinline__ float wsum4(const float (&s)[4], int lane) {
    const bool h32 = lane & 32, h16 = lane & 16;
    float t[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) t[i] = (h32 ? s[2 + i] : s[i]) + __shfl_xor(h32 ? s[i] : s[2 + i], 32);
    float w = (h16 ? t[1] : t[0]) + __shfl_xor(h16 ? t[0] : t[1], 16);
    w += __shfl_xor(w, 8);
    w += __shfl_xor(w, 4);
    w += __shfl_xor(w, 2);
    w += __shfl_xor(w, 1);
    return w;
}

// the wave's row sums of s[NR] -> st[row][wave]
template <int NR>
__device__ __forceinline__ void wave_row_sums(const float (&s)[NR], float (*st)[4], int lane, int wave) {
    if constexpr (NR % 8 == 0) {
#pragma unroll
        for (int h = 0; h < NR / 8; ++h) {
            float s8[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) s8[b] = s[8 * h + b];
            const float w = wsum8(s8, lane);
            if ((lane & 7) == 0) st[8 * h + ((lane >> 3) & 7)][wave] = w;
        }
    } else if constexpr (NR == 4) {
        const float w = wsum4(s, lane);
        if ((lane & 15) == 0) st[(lane >> 4) & 3][wave] = w;
    } else {
        static_assert(NR == 1, "row sums");
        float w = s[0];                                          // the plain butterfly in the same order of levels: the same tree
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
        if (lane == 0) st[0][wave] = w;
    }
}

// LayerNorm of NR rows of 768 (thread: columns 3 tid .. 3 tid + 2), two-pass statistics; rows 8 h .. 8 h + 7 go through one wsum8 each
template <int NR, class S>
__device__ __forceinline__ void ln8(float (&v)[NR][3], const float* __restrict__ g, const float* __restrict__ be, S& sm, int tid) {
    const int lane = tid & 63, wave = (tid >> 6) & 3;
    float gg[3], bb[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        gg[m] = GLOBAL_PTR(float, g)[3 * tid + m];
        bb[m] = GLOBAL_PTR(float, be)[3 * tid + m];
    }
    {
        float s[NR];
#pragma unroll
        for (int b = 0; b < NR; ++b) s[b] = (v[b][0] + v[b][1]) + v[b][2];
        wave_row_sums<NR>(s, sm.st1, lane, wave);
    }
    __syncthreads();
    float mean[NR];
#pragma unroll
    for (int b = 0; b < NR; ++b) {
        const float4 p = *reinterpret_cast<const float4*>(sm.st1[b]);
        mean[b] = ((p.x + p.y) + (p.z + p.w)) * (1.f / TC);
    }
    {
        float s[NR];
#pragma unroll
        for (int b = 0; b < NR; ++b) {
            const float d0 = v[b][0] - mean[b], d1 = v[b][1] - mean[b], d2 = v[b][2] - mean[b];
            s[b] = (d0 * d0 + d1 * d1) + d2 * d2;
        }
        wave_row_sums<NR>(s, sm.st2, lane, wave);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NR; ++b) {
        const float4 p = *reinterpret_cast<const float4*>(sm.st2[b]);
        const float rstd = rsqrtf(((p.x + p.y) + (p.z + p.w)) * (1.f / TC) + 1e-5f);
#pragma unroll
        for (int m = 0; m < 3; ++m) v[b][m] = (v[b][m] - mean[b]) * rstd * gg[m] + bb[m];
    }
}

// rows (thread: columns k = 3 tid + m) -> LDS activation tile
template <int NR, class S>
__device__ __forceinline__ void rows_to_tile(const float (&v)[NR][3], S& sm, int tid) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int k = 3 * tid + m;
        if constexpr (NR >= 4) {
#pragma unroll
            for (int q = 0; q < NR / 4; ++q)
                *reinterpret_cast<float4*>(sm.xs[q][k]) = make_float4(v[4 * q][m], v[4 * q + 1][m], v[4 * q + 2][m], v[4 * q + 3][m]);
        } else {
            sm.xs[0][k][0] = v[0][m];
        }
    }
}

// The weight / cache prefetches are loads of read-only memory with addresses known at the top of the layer: the compiler would hoist
// them all to there.  An opaque redefinition of the base pointer pins each prefetch to its place in the schedule.
template <class T>
__device__ __forceinline__ const T* pin_v(const T* p) {
    // (an opaque ZERO OFFSET, not an opaque pointer: the pointer keeps its global address space - a laundered pointer is generic, its
    // loads become FLAT instructions, which count on lgkmcnt as well and may complete out of order with the LDS reads around them)
    long long z = 0;
    asm volatile("" : "+v"(z)::"memory");
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + z);
}

template <int KT>
__device__ __forceinline__ void wload(float4 (&wr)[KT / 2], const float4* base, int tid, bool skip = false) {
    base = pin_v(base);
    if (skip) {                                    // DTTS_GPT_TOKEN_ABLATE bit 0 (measurement: the token's hops and hold without its weight bytes)
#pragma unroll
        for (int i = 0; i < KT / 2; ++i) wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
#pragma unroll
    for (int i = 0; i < KT / 2; ++i) wr[i] = ldg4(base + (i * 256 + tid));
}

// column GEMV on the LDS tile with the weight slice in registers (wr[i] = {W[k0][c], W[k0][c + 1], W[k1][c], W[k1][c + 1]},
// k0 = kl + KL 2 i, k1 = k0 + KL, c = 2 q): out[j] = result o = tid + 256 j of the NR x NC outputs, o = b * NC + col
template <int NR, int PN, int KL, int KT>
__device__ __forceinline__ void col_gemv(const float4 (&wr)[KT / 2], SmemT<NR>& sm, int tid, float (&out)[(NR * 2 * PN + 255) / 256]) {
    constexpr int NC = 2 * PN, NO = (NR * NC + 255) / 256;
    static_assert(KL * NR * NC <= Geo<NR>::RED && KT % 2 == 0, "col_gemv");
    const int q = tid % PN, kl = tid / PN;
    float acc[NR][2];
#pragma unroll
    for (int b = 0; b < NR; ++b) acc[b][0] = acc[b][1] = 0.f;
#pragma unroll
    for (int i = 0; i < KT / 2; ++i) {
        const int k0 = kl + KL * 2 * i, k1 = k0 + KL;
        const float4 w = wr[i];
        if constexpr (NR >= 4) {
#pragma unroll
            for (int rq = 0; rq < NR / 4; ++rq) {
                const float4 xa = *reinterpret_cast<const float4*>(sm.xs[rq][k0]), ya = *reinterpret_cast<const float4*>(sm.xs[rq][k1]);
                const float x[4] = {xa.x, xa.y, xa.z, xa.w};
                const float y[4] = {ya.x, ya.y, ya.z, ya.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int b = 4 * rq + e;
                    acc[b][0] += x[e] * w.x;
                    acc[b][1] += x[e] * w.y;
                    acc[b][0] += y[e] * w.z;
                    acc[b][1] += y[e] * w.w;
                }
            }
        } else {                                                 // one row: the same four FMAs per weight word, in the same order
            const float x = sm.xs[0][k0][0], y = sm.xs[0][k1][0];
            acc[0][0] += x * w.x;
            acc[0][1] += x * w.y;
            acc[0][0] += y * w.z;
            acc[0][1] += y * w.w;
        }
    }
    if (kl < KL) {
        float* r = sm.red + kl * (NR * NC) + q * 2;
#pragma unroll
        for (int b = 0; b < NR; ++b) *reinterpret_cast<float2*>(r + b * NC) = make_float2(acc[b][0], acc[b][1]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int o = tid + 256 * j;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (o < NR * NC) {
            const float* r = sm.red + o;
            int i = 0;
#pragma unroll 4
            for (; i + 4 <= KL; i += 4) {
                a0 += r[(i + 0) * (NR * NC)];
                a1 += r[(i + 1) * (NR * NC)];
                a2 += r[(i + 2) * (NR * NC)];
                a3 += r[(i + 3) * (NR * NC)];
            }
            for (; i < KL; ++i) a0 += r[i * (NR * NC)];
        }
        out[j] = (a0 + a1) + (a2 + a3);
    }
}

__device__ __forceinline__ float gelu_new(float v) {
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.f + tanhf(u));
}
