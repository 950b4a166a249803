// Split-precision (2 x fp16 planes, 3 products) conv GEMM for gfx950 - see conv_x3.h.
//
// 256 threads = 4 waves (2 x 2), block tile 128 x 192, wave tile 64 x 96 = 2 x 3 MFMA 32x32x16 tiles x 3 products = 18 MFMAs per
// K-step.  LDS: 2 W stages x 8 KiB + 2 X buffers x 12.1 KiB = 41 KiB -> 3 workgroups per CU.  Per K-step a wave issues 5 LDS-DMA
// loads (1 KiB each: 2 of the W tile, 3 of the X tile), 10 ds_read_b128 and 18 MFMAs behind ONE barrier.  Tile shape from
// tools/ubench/gemm_x3v.hip (M = 768 / 2304, B = 16, T = 936): 128 x 192 beats 128 x 128 by 8-10 % (2.5 instead of 8.6 % padded
// columns, 17 % fewer LDS-DMA bytes per MFMA); deeper pipelines and 256-row tiles measured within +-3 % of it.
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "conv_x3.h"
#include <cstdlib>
#include "prof.h"
#include "split3.h"

namespace dtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int NPL = XS_PLANES, NK = 2 * NPL;        // planes; "kinds" (plane, k-half) of a 16-channel K-step
constexpr int BM = 128, BN = X3_BN;
constexpr int WTILE = NK * BM * 16;                  // 8 KiB: [kind][128 rows][16 B]
constexpr int XMAIN = NK * BN * 16;                  // 12 KiB: [kind][192 columns][16 B]
constexpr int XBUF = XMAIN + NK * 2 * 16;            // + 2 halo columns per kind
// fused GroupNorm epilogue (EPI 3): exchange words are 16-byte {mean, M2, count, tag}, agent-scope (sc1) raw-buffer accesses
typedef unsigned gn_u4 __attribute__((ext_vector_type(4)));
constexpr int GN_AUX_SC1 = 16;                       // gfx940+ cache policy bit 4 = sc1
constexpr int GN_MAXW = 8;                           // (unused words area kept small)
constexpr int GN_SPIN_LIMIT = 1 << 22;
static_assert(7 * 32 <= 256 && 3 * GN_FUSE_MAX_NT <= 32 && (4 * 16 * 104 + 4 * GN_MAXW + 16 + 2 * BM) * 4 <= 2 * (WTILE + XBUF), "fused GroupNorm epilogue: LDS / lanes");

__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ wp, int C8, int CoutP, uint4* __restrict__ out) {
    const int m = blockIdx.x * 256 + threadIdx.x, c8 = blockIdx.y, tap = blockIdx.z;
    if (m >= CoutP) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = wp[((long long)tap * C8 * 8 + c8 * 8 + e) * CoutP + m] * XS_SCALE_W;
    uint4 q0, q1;
    split8(v, q0, q1);
    uint4* o = out + ((long long)(tap * C8 + c8) * NPL) * CoutP + m;
    o[0] = q0;
    o[CoutP] = q1;
}

template <int ACT, bool AB>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                          const float* __restrict__ ab, const int* __restrict__ lens, int T, int C8,
                                                          int Tp, uint4* __restrict__ out) {
    const int tp = blockIdx.x * 256 + threadIdx.x, c8 = blockIdx.y, b = blockIdx.z;
    if (tp >= Tp) return;
    const int t = tp - X3_HALO, len = lens ? lens[b] : T;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (t >= 0 && t < len) {
        const float* xr = x + (long long)b * x_bs + (long long)(c8 * 8) * x_cs + t;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xr[(long long)e * x_cs];
        if (AB) {
            const float* abr = ab + ((long long)b * C8 + c8) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = abr[2 * e] * v[e] + abr[2 * e + 1];
        }
        if (ACT == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * __frcp_rn(1.f + __expf(-v[e]));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= XS_SCALE_X;
        split8(v, q0, q1);
    }
    uint4* o = out + ((long long)(b * C8 + c8) * NPL) * Tp + tp;
    o[0] = q0;
    o[Tp] = q1;
}

// GroupNorm statistics + affine (+ AdaGN scale/shift) + activation + split in ONE kernel: a workgroup owns one (sample, group),
// reduces its cpg x len slab (shifted single-pass sums, as gn_coeffs_kernel), then re-reads the slab (L2-resident, <= 100 KB)
// and writes the group's cpg/8 chunk rows of the three planes.
template <int ACT>
__global__ __launch_bounds__(1024) void gn_split_planes_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                              const int* __restrict__ lens, int T, int C, int groups,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              const float* __restrict__ ada, int ada_stride, int ada_bs, int Tp,
                                                              uint4* __restrict__ out, const int* __restrict__ ada_idx) {
    __shared__ float red[2][16];
    __shared__ float sa[64], sd[64];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int len = lens ? lens[b] : T;
    const int cpg = C / groups;
    const float* xg = x + (long long)b * x_bs + (long long)(g * cpg) * x_cs;
    const float k = xg[0];
    float s1 = 0.f, s2 = 0.f;
    const bool vec = ((x_cs & 3) == 0) && ((reinterpret_cast<unsigned long long>(xg) & 15ull) == 0);
    if (vec) {
        const int len4 = len >> 2, n4 = cpg * len4;
        for (int i0 = tid; i0 < n4; i0 += 4 * nthr) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nthr, ic = i < n4 ? i : 0;
                const int row = ic / len4, t4 = ic - row * len4;
                v[u] = reinterpret_cast<const float4*>(xg + (long long)row * x_cs)[t4];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * nthr >= n4) continue;
                const float d0 = v[u].x - k, d1 = v[u].y - k, d2 = v[u].z - k, d3 = v[u].w - k;
                s1 += (d0 + d1) + (d2 + d3);
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        const int rem = len - (len4 << 2);
        for (int i = tid; i < cpg * rem; i += nthr) {
            const int row = i / rem, t = (len4 << 2) + (i - row * rem);
            const float d = xg[(long long)row * x_cs + t] - k;
            s1 += d;
            s2 += d * d;
        }
    } else {
        for (int i = tid; i < cpg * len; i += nthr) {
            const int row = i / len, t = i - row * len;
            const float d = xg[(long long)row * x_cs + t] - k;
            s1 += d;
            s2 += d * d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
    for (int i = 0; i < (nthr >> 6); ++i) { S1 += red[0][i]; S2 += red[1][i]; }
    const float n = (float)(cpg * len);
    S1 /= n;
    S2 /= n;
    const float mean = k + S1, var = fmaxf(S2 - S1 * S1, 0.f), rstd = rsqrtf(var + eps);
    if (tid < cpg) {
        const int c = g * cpg + tid;
        float a = rstd * gamma[c];
        float d = beta[c] - mean * a;
        if (ada) {
            const float* ad = ada + (ada_idx ? (long long)ada_idx[b] : (long long)b * ada_bs);
            const float sc = 1.f + ad[(long long)c * ada_stride], sh = ad[(long long)(C + c) * ada_stride];
            a *= sc;
            d = d * sc + sh;
        }
        sa[tid] = a;
        sd[tid] = d;
    }
    __syncthreads();
    const int c8n = cpg >> 3, C8 = C >> 3;
    // every workgroup of a (sample, group) computes the statistics (the slab is L2-resident); the split work is divided among them
    for (int item = blockIdx.z * nthr + tid; item < c8n * Tp; item += nthr * gridDim.z) {
        const int c8l = item / Tp, tp = item - c8l * Tp, t = tp - X3_HALO;
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
        if (t >= 0 && t < len) {
            const float* xr = xg + (long long)(c8l * 8) * x_cs + t;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xr[(long long)e * x_cs];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = sa[c8l * 8 + e] * v[e] + sd[c8l * 8 + e];
                if (ACT == ACT_SILU) v[e] = v[e] * __frcp_rn(1.f + __expf(-v[e]));
                v[e] *= XS_SCALE_X;
            }
            split8(v, q0, q1);
        }
        uint4* o = out + ((long long)(b * C8 + g * c8n + c8l) * NPL) * Tp + tp;
        o[0] = q0;
        o[Tp] = q1;
    }
}

// The same with the slab held in REGISTERS between the two phases (one HBM read): thread <-> NI (8-channel chunk, column) items of the
// output, loaded as 8 row-strided floats each (a wave reads 256 contiguous bytes per row), reduced, then normalised and split where
// they sit.  Covers cpg/8 x Tp <= NI x 1024 items (T <= 1022 at cpg = 24); longer sequences take the two-pass kernel above.
template <int ACT, int NI>
__global__ __launch_bounds__(1024) void gn_split_planes_reg_kernel(const float* __restrict__ x, long long x_bs, int x_cs,
                                                                  const int* __restrict__ lens, int T, int C, int groups,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                  const float* __restrict__ ada, int ada_stride, int ada_bs, int Tp,
                                                                  uint4* __restrict__ out, const int* __restrict__ ada_idx) {
    __shared__ float red[2][16];
    __shared__ float sa[64], sd[64];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = lens ? lens[b] : T;
    const int cpg = C / groups, c8n = cpg >> 3, C8 = C >> 3, nitems = c8n * Tp;
    const float* xg = x + (long long)b * x_bs + (long long)(g * cpg) * x_cs;
    const float k = xg[0];
    float v[NI][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int item = tid + q * 1024, c8l = item / Tp, t = item - c8l * Tp - X3_HALO;
        const bool ok = item < nitems && t >= 0 && t < len;
        const float* xr = xg + (long long)(ok ? c8l * 8 : 0) * x_cs + (ok ? t : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[q][e] = xr[(long long)e * x_cs];
        if (ok) {
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = v[q][e] - k;
            s1 += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
            s2 += ((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) + ((d[4] * d[4] + d[5] * d[5]) + (d[6] * d[6] + d[7] * d[7]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    float S1 = 0.f, S2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { S1 += red[0][i]; S2 += red[1][i]; }
    const float n = (float)(cpg * len);
    S1 /= n;
    S2 /= n;
    const float mean = k + S1, var = fmaxf(S2 - S1 * S1, 0.f), rstd = rsqrtf(var + eps);
    if (tid < cpg) {
        const int c = g * cpg + tid;
        float a = rstd * gamma[c];
        float d = beta[c] - mean * a;
        if (ada) {
            const float* ad = ada + (ada_idx ? (long long)ada_idx[b] : (long long)b * ada_bs);
            const float sc = 1.f + ad[(long long)c * ada_stride], sh = ad[(long long)(C + c) * ada_stride];
            a *= sc;
            d = d * sc + sh;
        }
        sa[tid] = a;
        sd[tid] = d;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int item = tid + q * 1024;
        if (item >= nitems) continue;
        const int c8l = item / Tp, tp = item - c8l * Tp, t = tp - X3_HALO;
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
        if (t >= 0 && t < len) {
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                w[e] = sa[c8l * 8 + e] * v[q][e] + sd[c8l * 8 + e];
                if (ACT == ACT_SILU) w[e] = w[e] * __frcp_rn(1.f + __expf(-w[e]));
                w[e] *= XS_SCALE_X;
            }
            split8(w, q0, q1);
        }
        uint4* o = out + ((long long)(b * C8 + g * c8n + c8l) * NPL) * Tp + tp;
        o[0] = q0;
        o[Tp] = q1;
    }
}

// EPI 0: bias (+ residual); 1: + activation / out_scale.   KW3: three taps (else one).  NSTG: LDS stages (2: every K-step waits for
// the loads issued during the previous one; 3: loads run two steps ahead and the wait is a counted vmcnt).
// K loop order is (16-channel block, tap): the X tile of a channel block carries its halo (192 + KW - 1 columns) and is fetched
// ONCE, every tap reads it through a shifted (16-byte aligned) ds_read_b128; only the W tile changes per tap.
// Every wave issues the SAME number of LDS-DMA instructions per step (5 for k = 1; 3, + 1 halo at the last tap, for k = 3), so the
// counted wait is one immediate for all waves; VMEM loads complete in order.
template <int EPI, bool KW3, int NSTG>
__global__ __launch_bounds__(256, (NSTG == 2 && EPI != 3) ? 3 : 2) void conv_x3_kernel(ConvParams p) {
    constexpr int KW = KW3 ? 3 : 1, D = NSTG - 1;       // D: prefetch distance in steps (W, X of k = 1) / channel blocks (X of k = 3)
    constexpr int XOFF = NSTG * WTILE;                  // LDS: W stages | X buffers | bias
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    // 1-D grid, XCD-aware: the M tiles of one (sample, N tile) are adjacent logical ids -> they share the X tile in one L2
    const int mtiles = p.CoutP / BM, ntiles = (p.Nout + BN - 1) / BN;
    const int Ls = xcd_remap(blockIdx.x, gridDim.x);
    const int S = p.ksplit, L = Ls / S, z = Ls - L * S;                  // the splits of a tile are adjacent logical ids (one XCD)
    // EPI 3 (fused GroupNorm): (sample, M tile, N tile) order - a tile waits for statistics of tiles at most 2 N - 1 ids ahead (conv_x3.h)
    const bool mt_major = EPI == 3 && !(p.ablate & 16);
    int mt = mt_major ? (L / ntiles) % mtiles : L % mtiles;
    int b = mt_major ? L / (ntiles * mtiles) : (L / mtiles) / ntiles;
    int nti = mt_major ? L % ntiles : (L / mtiles) - b * ntiles;
    int bn = L / mtiles;                                                 // (sample, N tile) column of this tile: index into p.cols
    if (EPI == 2 && mtiles > 6 && mtiles % 6 == 0 && !(p.ablate & 1024)) {
        // Tall launches (the qkv conv: 18 M tiles): with the M tiles of an X tile adjacent, the ~30 workgroups an XCD runs at a time
        // stream ALL 18 weight tiles (7 MB through a 4 MB L2) for 1.7 X tiles - and again in the next round: 138 MB fetched per launch
        // for 30 MB of operands.  Order the ids of a block of 5 (sample, N tile) columns as [M group of 6][column][M tile in the group]
        // instead: a round of 30 = 6 weight tiles (2.4 MB, each shared by 5 workgroups) x 5 X tiles (2.9 MB, each shared by 6).
        const int nbn = p.cols ? p.ncols : ntiles * p.B, per_blk = 5 * mtiles, blk = L / per_blk, r = L - blk * per_blk;
        const int bn0 = blk * 5, nb = min(5, nbn - bn0), per_grp = nb * 6, mg = r / per_grp, r2 = r - mg * per_grp, bnl = r2 / 6;
        mt = mg * 6 + (r2 - bnl * 6);
        bn = bn0 + bnl;
        b = bn / ntiles;
        nti = bn - b * ntiles;
    }
    if (p.cols) {                                                        // ragged batch: live columns only (never with EPI 3)
        const int pk = __builtin_amdgcn_readfirstlane(p.cols[bn]);
        b = (pk >> 8) - p.cols_b0;
        nti = pk & 255;
    }
    const int m0 = mt * BM, n0 = nti * BN;
    const int nvalid = p.len_out ? p.len_out[b] : p.Nout;
    if (n0 >= nvalid) return;
    const int C8 = p.Cin >> 3, call = p.Cin >> 4, Tp = p.x3_tp;
    // split z owns channel blocks [cbeg, cend): the loop below indexes them from 0 through the shifted base pointers
    const int cbeg = (int)((long long)call * z / S), c16n = (int)((long long)call * (z + 1) / S) - cbeg, nks = KW * c16n;
    const int bin = p.x_bidx ? p.x_bidx[b] : b;
    // ---- operand addressing: UNIFORM byte pointers (SGPR pairs) + one per-lane 32-bit offset (lane * 16), so that every LDS-DMA piece is
    // `global_load_lds_dwordx4 v_lane16, s[ptr]`: no per-piece VALU address arithmetic, and the pointers advance by constant strides
    // (round 2 recomputed each piece's address from (step, kind, half) - 45 SALU + 12 VALU per K-step against 18 MFMAs).
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned char* wb8 = static_cast<const unsigned char*>(p.w3) + ((long long)m0 + (long long)(2 * cbeg) * NPL * p.CoutP) * 16;
    const unsigned char* xb8 = static_cast<const unsigned char*>(p.x3) +
                               ((long long)bin * C8 * NPL * Tp + n0 + (X3_HALO - p.pad) + (long long)(2 * cbeg) * NPL * Tp) * 16;
    const long long wtapB = (long long)C8 * NPL * p.CoutP * 16;           // bytes between taps
    const long long wblkB = (long long)2 * NPL * p.CoutP * 16;            // bytes between 16-channel blocks (W)
    const long long xblkB = (long long)2 * NPL * Tp * 16;                 //                              (X)
    auto dma = [&](const unsigned char* g, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane16),
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    // This wave's pieces (1 KiB = 64 rows / columns of one (plane, k-half) "kind").  W: kind = wave, row halves 0 / 1.  X (k = 1): kind = wave,
    // column blocks 0..2.  X (k = 3): the 12 pieces of a channel block are spread over its three taps, piece tap * 4 + wave at tap `tap`.
    const int wkind = wave, wpl = wkind >> 1, wh = wkind & 1;
    const unsigned char* wq = wb8 + (long long)(wh * NPL + wpl) * p.CoutP * 16;          // W piece pointer of the step being ISSUED (row half 0)
    const int wlds = wkind * (BM * 16);                                                   // + stage * WTILE (+ 1024 for row half 1)
    const unsigned char* xq = xb8 + (long long)(wh * NPL + wpl) * Tp * 16;               // k = 1: X pieces of the step being issued (column block 0)
    const int xlds = XOFF + wkind * (BN * 16);
    // k = 3: per-tap (kind, column block) of this wave's X piece
    long long x3o[3];
    int x3l[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int j = t * 4 + wave, kind = j / 3, cb = j - kind * 3;
        x3o[t] = ((long long)((kind & 1) * NPL + (kind >> 1)) * Tp + cb * 64) * 16;
        x3l[t] = XOFF + kind * (BN * 16) + cb * 1024;
    }
    // k = 3 halo: lanes 0..7 fetch (kind = lane >> 1, column 192 + (lane & 1)); expressed against the lane16 offset every piece carries
    const long long halo_lane = (lane < 2 * NK) ? (((long long)(((lane >> 1) & 1) * NPL + (lane >> 2)) * Tp + BN + (lane & 1)) * 16 - (long long)lane16) : 0;
    const unsigned char* xblk = xb8;                                                      // k = 3: block pointer of the X tile being issued
    auto issue_w = [&](int i, int stage) { dma(wq + i * 1024, stage * WTILE + wlds + i * 1024); };
    auto issue_x1 = [&](int i, int stage) { dma(xq + i * 1024, stage * XBUF + xlds + i * 1024); };
    auto issue_x3 = [&](int t, int stage) { dma(xblk + x3o[t], stage * XBUF + x3l[t]); };
    auto issue_halo = [&](int stage) {
        if (lane < 2 * NK)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xblk + halo_lane + lane16),
                                             (__attribute__((address_space(3))) void*)(smem + XOFF + stage * XBUF + XMAIN), 16, 0, 0);
    };
    // advance the W pointer by one step: next tap, or the first tap of the next channel block
    int itap = 0;                                       // tap of the step being issued (k = 3)
    auto next_w = [&]() {
        if (!KW3) wq += wblkB;
        else if (itap < 2) { wq += wtapB; ++itap; }
        else { wq += wblkB - 2 * wtapB; itap = 0; }
    };

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 96;
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue, oldest data first: W of steps 0 .. D-1; X of steps 0 .. D-1 (k = 1) / of channel blocks 0 .. D-1 (k = 3).  Steps / blocks
    // beyond the end are not issued at all; the counted waits below fall back to a full drain for the last D - 1 steps.
#pragma unroll
    for (int q = 0; q < D; ++q) {
        if (q < nks) {
            issue_w(0, q);
            issue_w(1, q);
        }
        next_w();
        if (!KW3) {
            if (q < nks) {
#pragma unroll
                for (int i = 0; i < 3; ++i) issue_x1(i, q);
            }
            xq += xblkB;
        } else {
            if (q < c16n) {
#pragma unroll
                for (int t = 0; t < 3; ++t) issue_x3(t, q);
                issue_halo(q);
            }
            xblk += xblkB;
        }
    }
    // the tile's 128 bias values -> LDS (read back in the epilogue; the first K-step barrier orders the write)
    float* bias_s = reinterpret_cast<float*>(smem + XOFF + NSTG * XBUF);
    if (tid < BM) bias_s[tid] = (p.bias && m0 + tid < p.Cout) ? p.bias[m0 + tid] : 0.f;
    // fused GroupNorm: the norm's per-row affine with the AdaGN (1 + scale, shift) folded in -> LDS now, so that the kernel's tail has
    // no global round trip left but the statistics exchange itself:  y_hat = (y - mu) rstd ga + gb
    float* ga_s = bias_s + BM + 4;
    float* gb_s = ga_s + BM;
    if (EPI == 3 && tid < BM) {
        const int c = m0 + tid;
        float ga = p.gn_gamma[c], gb = p.gn_beta[c];
        if (p.gn_ada) {
            const float* ad = p.gn_ada + (p.gn_ada_idx ? (long long)p.gn_ada_idx[b] : 0);
            const float sc = 1.f + ad[(long long)c * p.gn_ada_stride], sh = ad[(long long)(p.Cout + c) * p.gn_ada_stride];
            ga *= sc;
            gb = gb * sc + sh;
        }
        ga_s[tid] = ga;
        gb_s[tid] = gb;
    }

    int c16 = 0, tap = 0;                               // the step being computed
    int sw = 0, sx = 0;                                 // its W stage / X buffer; the step being issued uses (sw + D) % NSTG, (sx + D) % NSTG
    // MFMA fragments of one K-step <- LDS
    auto read_frags = [&](hf8 (&fa)[2][NPL], hf8 (&fb)[3][NPL], int stw, int stx, int tp) {
        const unsigned char* As = smem + stw * WTILE + lhi * (BM * 16);
        const unsigned char* Xb = smem + XOFF + stx * XBUF;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int nn = wn0 + j * 32 + l31 + tp;                     // column of the haloed tile
            const unsigned char* xr = nn < BN ? Xb + lhi * (BN * 16) + nn * 16 : Xb + XMAIN + lhi * 32 + (nn - BN) * 16;
            const int xps = nn < BN ? 2 * BN * 16 : 64;                 // plane stride: main part / halo part
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fb[j][pl] = *reinterpret_cast<const hf8*>(xr + pl * xps);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[i][pl] = *reinterpret_cast<const hf8*>(As + pl * (2 * BM * 16) + (wm0 + i * 32 + l31) * 16);
    };
    // one LDS-DMA piece of the step being issued (slot 0..4 of the K-step: after the first five MFMA triples)
    auto issue_slot = [&](int slot, int swi, int sxi, bool wlive, bool xlive) {
        __builtin_amdgcn_sched_barrier(0);
        if (slot < 2) { if (wlive) issue_w(slot, swi); }
        else if (!KW3) { if (xlive) issue_x1(slot - 2, sxi); }
        else if (slot == 2) { if (xlive) issue_x3(tap, sxi); }
        else if (slot == 3 && tap == 2) { if (xlive) issue_halo(sxi); }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto advance = [&]() {
        next_w();
        if (!KW3) xq += xblkB;
        sw = sw + 1 == NSTG ? 0 : sw + 1;
        if (++tap == KW) {
            tap = 0;
            ++c16;
            sx = sx + 1 == NSTG ? 0 : sx + 1;
            if (KW3) xblk += xblkB;
        }
    };
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};   // term-major: one cross product over the wave's 6 accumulators per group, smallest terms first
    {
    hf8 a[2][NPL], bb[3][NPL];
    int ks = 0;
    // ---- main part, unrolled over one PERIOD of the (W stage, X buffer, tap) cycle: NSTG steps for k = 1, 3 NSTG for k = 3.  Inside it
    // every stage index, tap, LDS offset and the form of the pointer advance is a compile-time constant and every piece is live, so
    // a K-step carries its 18 MFMAs, 10 fragment reads, 5 pieces and a handful of scalar pointer adds - no stage / tap / liveness
    // bookkeeping (round 3: 3.7 scalar instructions per MFMA, and with one wave per SIMD the wave's own issue slots between two MFMAs
    // are what the K-step runs out of).  The generic loop below finishes the last < PERIOD + D steps (and runs everything when
    // DTTS_CONV_UNROLL=0: p.ablate bit 9).
    constexpr int PERIOD = KW * NSTG;
    auto ustep = [&](auto kc) {
        constexpr int k = decltype(kc)::value;                        // step index inside the period
        constexpr int TAPc = k % KW, SWc = k % NSTG, SXc = (k / KW) % NSTG;
        constexpr int SWI = (SWc + D) % NSTG, SXI = (SXc + D) % NSTG;
        if (NSTG == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (KW3 ? 3 : 5)) : "memory");
        __builtin_amdgcn_s_barrier();
        constexpr bool ASM_READS = NSTG >= 3;          // (the 2-stage instantiations live on 168 VGPRs: the wider live ranges would spill)
        if (!ASM_READS) read_frags(a, bb, SWc, SXc, TAPc);
        else {
            // Fragments in the order the MFMA groups below consume them (low-plane A 0 + high-plane X 0..2 | low A 1 | high A 0 + low
            // X 0..2 | high A 1), read by hand-written ds_read_b128 with COUNTED lgkmcnt waits in front of each group: the first group
            // starts when 4 of the 10 reads have landed.  (Compiler-issued reads always get `s_waitcnt lgkmcnt(0)` in front of the first
            // MFMA in this loop - the LDS-DMA traffic makes its wait insertion conservative - and with one wave per SIMD nothing else
            // covers the ~10 x 1 KiB of LDS return time.)  The waits carry the fragment registers as in/out operands, so no use can be
            // scheduled in front of its wait; an in-flight MFMA has read its operands long before a read issued after it returns.
            const unsigned as_addr = (unsigned)(SWc * WTILE) + lhi * (BM * 16) + (wm0 + l31) * 16;
            const unsigned xb_addr = (unsigned)(XOFF + SXc * XBUF);
            auto rd = [&](hf8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); };
            auto a_addr = [&](int i, int pl) { return as_addr + pl * (2 * BM * 16) + i * 32 * 16; };
            auto x_addr = [&](int j, int pl) {
                const int nn = wn0 + j * 32 + l31 + TAPc;
                return nn < BN ? xb_addr + lhi * (BN * 16) + nn * 16 + pl * (2 * BN * 16) : xb_addr + XMAIN + lhi * 32 + (nn - BN) * 16 + pl * 64;
            };
            rd(a[0][1], a_addr(0, 1)); rd(bb[0][0], x_addr(0, 0)); rd(bb[1][0], x_addr(1, 0)); rd(bb[2][0], x_addr(2, 0));
            rd(a[1][1], a_addr(1, 1));
            rd(a[0][0], a_addr(0, 0)); rd(bb[0][1], x_addr(0, 1)); rd(bb[1][1], x_addr(1, 1)); rd(bb[2][1], x_addr(2, 1));
            rd(a[1][0], a_addr(1, 0));
        }
        int slot = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ASM_READS && t == 0 && i == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a[0][1]), "+v"(bb[0][0]), "+v"(bb[1][0]), "+v"(bb[2][0]));
                if (ASM_READS && t == 0 && i == 1) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(a[1][1]));
                if (ASM_READS && t == 1 && i == 0) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[0][0]), "+v"(bb[0][1]), "+v"(bb[1][1]), "+v"(bb[2][1]));
                if (ASM_READS && t == 1 && i == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1][0]));
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
                if (slot < 5) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (slot < 2) issue_w(slot, SWI);
                    else if (!KW3) issue_x1(slot - 2, SXI);
                    else if (slot == 2) issue_x3(TAPc, SXI);
                    else if (slot == 3 && TAPc == 2) issue_halo(SXI);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ++slot;
            }
        // pointers of the step issued next: W one tap / one channel block further (the tap being ISSUED is (TAPc + D) % KW), X per block
        if (!KW3) {
            wq += wblkB;
            xq += xblkB;
        } else {
            if ((TAPc + D) % KW < 2) wq += wtapB;
            else wq += wblkB - 2 * wtapB;
            if (TAPc == KW - 1) xblk += xblkB;
        }
    };
    auto uperiod = [&](auto... kc) { (ustep(kc), ...); };
    // measured at the bench's 240-tile launches (tools/bench_forward.py): k = 3 85.3 -> 75.6 us (- 11 %, of which the hand-written
    // fragment reads 1.5), qkv conv 75.7 -> 75.0; the 1 x 1 convs of the ResBlock / proj 34.8 -> 36.7 (+ 5 %: five pieces per step -
    // they keep the generic loop)
    constexpr bool UNROLL = KW3 || EPI == 2;
    if (UNROLL && !(p.ablate & 512)) {
        // whole periods while every step of the period (and the D steps it issues ahead) lies inside the loop
        // (k = 3: the X block of channel block c16 + D is issued piece by piece over c16's three taps, so c16 + D must exist too)
        const int nper = KW3 ? (c16n - D) / NSTG : (nks - D) / PERIOD;
        for (int q = 0; q < nper; ++q) {
            if constexpr (PERIOD == 2) uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            else if constexpr (PERIOD == 3) uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            else if constexpr (PERIOD == 4) uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
            else if constexpr (PERIOD == 6) uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
            else if constexpr (PERIOD == 9) uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{});
            else uperiod(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{});
        }
        // hand over to the generic loop: a whole number of periods leaves stage, buffer and tap where they started
        ks = nper * PERIOD;
        c16 = ks / KW;
        itap = D % KW;
    }
    for (; ks < nks; ++ks) {
        // Data of this step was issued D steps (blocks) ago; the loads of the D - 1 steps issued since may stay in flight.  The
        // barrier also orders the previous step's ds_reads of the stage / buffer refilled next (WAR).
        // counted wait only while every one of the last D - 1 iterations issued its full set (2 W + 1 X for k = 3, 2 + 3 for k = 1)
        if (NSTG == 2 || ks + D > nks || (KW3 && c16 + D >= c16n)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (KW3 ? 3 : 5)) : "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(a, bb, sw, sx, tap);
        // ONE LDS-DMA piece after every three MFMAs (as a burst the pieces of a CU's waves queue on the texture-address path while
        // every MFMA pipe idles and the co-resident workgroups fall into lock-step)
        const int swi = sw + D >= NSTG ? sw + D - NSTG : sw + D, sxi = sx + D >= NSTG ? sx + D - NSTG : sx + D;
        const bool wlive = ks + D < nks, xlive = KW3 ? c16 + D < c16n : ks + D < nks;
        int slot = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
                if (slot < 5) issue_slot(slot, swi, sxi, wlive, xlive);
                ++slot;
            }
        advance();
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (S > 1) {
        // raw accumulators -> this split's slab (lane-contiguous: one 256-byte run per register and wave); the last workgroup of the tile
        // to arrive adds the S slabs in split order (its own included, from memory: the order never depends on who arrives last)
        // Every slab access is an agent-scope (sc1) access: coherent across the XCDs' L2s one by one, WITHOUT the bulk L2 write-back /
        // invalidate a release / acquire fence costs here (measured: +40 us per launch).
        float* slab = p.kpart + ((size_t)L * S) * (96 * 256);
        float* mine = slab + (size_t)z * (96 * 256) + tid;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + ((i * 3 + j) * 16 + r) * 256, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's stores have reached the coherent level
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(bias_s + BM);      // dynamic LDS (a second __shared__ object would cost the K loop its counted waits)
        if (tid == 0) *s_last = __hip_atomic_fetch_add(p.kcount + L, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1 ? 1 : 0;
        __syncthreads();
        if (!*s_last) return;
        if (tid == 0) __hip_atomic_store(p.kcount + L, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int zz = 0; zz < S; ++zz) {
            const float* src = slab + (size_t)zz * (96 * 256) + tid;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][j][r] += __hip_atomic_load(src + ((i * 3 + j) * 16 + r) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    if (EPI == 3) {
        // ---- fused GroupNorm epilogue (conv_x3.h).  Phases: (1) y = acc * s + bias (+ residual, transposed through the LDS so that its
        // rows arrive 16 B per lane); (2) two-pass statistics of the wave's 8 row chunks x 96 columns -> published as tagged words;
        // (3) the fp32 rows of y, if anybody needs them (16-byte stores through the LDS) - issued BEFORE the poll so that the neighbours'
        // latency is spent on useful stores; (4) poll the words of the tile's groups, combine, per-row coefficients; (5) normalise,
        // activate, split and store the planes from the registers (a lane pair exchanges half chunks: v_permlane32_swap).
        constexpr int EP_LD = 104;
        if (p.ablate & 64) return;
        __syncthreads();                                               // every wave has left the K loop: its LDS stages are free
        float* st = reinterpret_cast<float*>(smem) + wave * (16 * EP_LD);
        float* prt = reinterpret_cast<float*>(smem) + 4 * 16 * EP_LD;   // [word][4]: mean, M2, count of one (chunk, N tile, column half)
        float* gst = prt + 4 * GN_MAXW;                                 // [8 groups][2]: mean, rstd
        float* sa = gst + 16;                                           // [128 rows]: y -> a y + d
        float* sd = sa + BM;
        const int ncol0 = n0 + wn0;
        const float* rb = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
        // every residual load of the tile is in flight before the first one is used: one memory latency instead of four
        float4 rv[4][6];
        if (rb) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = lane + 64 * k, rl = idx / 24, c4 = idx - rl * 24, row = m0 + wm0 + (q >> 1) * 32 + (q & 1) * 16 + rl, n = ncol0 + c4 * 4;
                    const float* src = rb + (long long)row * p.res_cs + n;
                    rv[q][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n + 3 < nvalid) rv[q][k] = *reinterpret_cast<const float4*>(src);
                    else if (n < nvalid) {
                        rv[q][k].x = src[0];
                        if (n + 1 < nvalid) rv[q][k].y = src[1];
                        if (n + 2 < nvalid) rv[q][k].z = src[2];
                    }
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q >> 1, rowt0 = wm0 + i * 32 + (q & 1) * 16;
            if (rb) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = lane + 64 * k, rl = idx / 24, c4 = idx - rl * 24;
                    *reinterpret_cast<float4*>(st + rl * EP_LD + c4 * 4) = rv[q][k];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = (q & 1) * 8 + rr, rl = (rr & 3) + 8 * (rr >> 2) + 4 * lhi;
                    float v = acc[i][j][r] * XS_ACC_SCALE + bias_s[rowt0 + rl];
                    if (rb) v += p.res_scale * st[rl * EP_LD + j * 32 + l31];
                    acc[i][j][r] = v;
                }
            if (rb) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- (2) statistics of chunk ck = i * 4 + (r >> 2) (8 rows x this wave's valid columns), two passes in registers
        bool okj[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) okj[j] = ncol0 + j * 32 + l31 < nvalid;
        const int nvh = min(max(nvalid - ncol0, 0), 96);
        const float cnt = 8.f * (float)nvh, rcnt = nvh > 0 ? 1.f / cnt : 0.f;
        float s8[8];
#pragma unroll
        for (int ck = 0; ck < 8; ++ck) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) a += okj[j] ? acc[ck >> 2][j][4 * (ck & 3) + e] : 0.f;
            s8[ck] = a;
        }
        const float tot = wave_sum8(s8, lane);
#pragma unroll
        for (int ck = 0; ck < 8; ++ck) {
            const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 8 * ck)) * rcnt;
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dv = acc[ck >> 2][j][4 * (ck & 3) + e] - mu;
                    a += okj[j] ? dv * dv : 0.f;
                }
            s8[ck] = a;
        }
        const float m2 = wave_sum8(s8, lane);
        const int C8o = p.Cout >> 3, NTs = ntiles;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.gn_xch, (short)0, (int)((size_t)p.B * C8o * NTs * 2 * 16), 0x00020000);
        if ((lane & 7) == 0) {
            const int chunk = ((m0 + wm0) >> 3) + (lane >> 3);
            const gn_u4 wv = {__float_as_uint(tot * rcnt), __float_as_uint(m2), __float_as_uint(cnt), p.gn_tag};
            __builtin_amdgcn_raw_buffer_store_b128(wv, xrs, (((b * C8o + chunk) * NTs + nti) * 2 + (wave & 1)) * 16, 0, GN_AUX_SC1);
        }
        // ---- (3) fp32 rows of y (residual stream), 16 bytes per lane through the LDS
        if (p.y) {
            float* yb = p.y + (long long)b * p.y_bs;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = q >> 1, rowt0 = wm0 + i * 32 + (q & 1) * 16;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = (q & 1) * 8 + rr, rl = (rr & 3) + 8 * (rr >> 2) + 4 * lhi;
                        st[rl * EP_LD + j * 32 + l31] = acc[i][j][r];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int idx = lane + 64 * k, rl = idx / 24, c4 = idx - rl * 24, row = m0 + rowt0 + rl, n = ncol0 + c4 * 4;
                    const float4 a4 = *reinterpret_cast<const float4*>(st + rl * EP_LD + c4 * 4);
                    if (n < nvalid) {
                        float* dst = yb + (long long)row * p.y_cs + n;
                        if (n + 3 < nvalid) *reinterpret_cast<float4*>(dst) = a4;
                        else {
                            dst[0] = a4.x;
                            if (n + 1 < nvalid) dst[1] = a4.y;
                            if (n + 2 < nvalid) dst[2] = a4.z;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (p.ablate & 128) return;
        // ---- (4) the statistics of the groups this tile's rows belong to.  Thread (group slot, k): k = (chunk of the group, valid N
        // tile) polls the TWO column-half words of that (chunk, N tile) and combines them; a group's 3 x nvt pairs sit in one 16- or
        // 32-lane segment of a wave and are combined there with xor shuffles (Chan's parallel variance, fixed order: deterministic)
        const int cpg8 = (p.Cout / p.gn_groups) >> 3;                                  // chunks per group (3)
        const int ct0 = m0 >> 3, g_lo = ct0 / cpg8, g_hi = (ct0 + BM / 8 - 1) / cpg8;
        const int nvt = (nvalid + BN - 1) / BN, ppg = cpg8 * nvt;                      // valid N tiles; pairs per group (<= 18)
        const int SL = ppg <= 16 ? 16 : 32;
        {
            const int gs = tid / SL, k = tid - gs * SL;
            const bool active = gs <= g_hi - g_lo && k < ppg;
            float pn = 0.f, pm = 0.f, pq = 0.f;
            if (active) {
                const int ch = k / nvt, nt = k - ch * nvt;
                const int off = (((b * C8o + (g_lo + gs) * cpg8 + ch) * NTs + nt) * 2) * 16;
                gn_u4 v0, v1;
                int spins = 0;
                for (;;) {
                    v0 = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, GN_AUX_SC1);
                    v1 = __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16, 0, GN_AUX_SC1);
                    if ((v0.w == p.gn_tag && v1.w == p.gn_tag) || (p.ablate & 8)) break;
                    if (++spins > GN_SPIN_LIMIT) {
                        __hip_atomic_store(p.gn_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float m0v = __uint_as_float(v0.x), q0v = __uint_as_float(v0.y), n0v = __uint_as_float(v0.z);
                const float m1v = __uint_as_float(v1.x), q1v = __uint_as_float(v1.y), n1v = __uint_as_float(v1.z);
                pn = n0v + n1v;
                const float rn = pn > 0.f ? 1.f / pn : 0.f, dm = m0v - m1v;
                pm = (n0v * m0v + n1v * m1v) * rn;
                pq = q0v + q1v + n0v * n1v * rn * dm * dm;
            }
            float N = pn, S = pn * pm;
            for (int o = 1; o < SL; o <<= 1) {
                N += __shfl_xor(N, o);
                S += __shfl_xor(S, o);
            }
            const float mu = N > 0.f ? S / N : 0.f, dmu = pm - mu;
            float M2 = pq + pn * dmu * dmu;
            for (int o = 1; o < SL; o <<= 1) M2 += __shfl_xor(M2, o);
            if (active && k == 0) {
                gst[2 * gs] = mu;
                gst[2 * gs + 1] = rsqrtf((N > 0.f ? M2 / N : 0.f) + p.gn_eps);
            }
        }
        __syncthreads();
        if (tid < BM) {                                                                // y_hat = a y + d per row
            const int g = (m0 + tid) / (p.Cout / p.gn_groups) - g_lo;
            const float a = gst[2 * g + 1] * ga_s[tid];
            sa[tid] = a;
            sd[tid] = gb_s[tid] - gst[2 * g] * a;
        }
        __syncthreads();
        if (p.ablate & 256) return;
        // ---- (5) normalise + activation + split -> the consumer's planes.  Columns of the tile beyond the sample's length are
        // written as zeros (a k = 3 consumer reads one column past the end), and so are the halo columns next to the valid range.
        const int Tp = p.x3_tp;
        unsigned char* ob = static_cast<unsigned char*>(p.gn_out3) + (size_t)b * C8o * NPL * Tp * 16;
        const bool silu = p.gn_act == ACT_SILU;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = ncol0 + j * 32 + l31;
                const bool ok = n < nvalid;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int r16 = wm0 + i * 32 + 16 * k2;
                    float ve[4], vo[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ra = r16 + e + 4 * lhi, rb2 = ra + 8;
                        float x0 = acc[i][j][8 * k2 + e] * sa[ra] + sd[ra], x1 = acc[i][j][8 * k2 + 4 + e] * sa[rb2] + sd[rb2];
                        if (silu) {
                            x0 = x0 * __frcp_rn(1.f + __expf(-x0));
                            x1 = x1 * __frcp_rn(1.f + __expf(-x1));
                        }
                        ve[e] = ok ? x0 * XS_SCALE_X : 0.f;
                        vo[e] = ok ? x1 * XS_SCALE_X : 0.f;
                    }
                    unsigned we0[2], we1[2], wo0[2], wo1[2];
                    split_pair(ve[0], ve[1], we0[0], we1[0]);
                    split_pair(ve[2], ve[3], we0[1], we1[1]);
                    split_pair(vo[0], vo[1], wo0[0], wo1[0]);
                    split_pair(vo[2], vo[3], wo0[1], wo1[1]);
                    const auto s00 = __builtin_amdgcn_permlane32_swap(we0[0], wo0[0], false, false);
                    const auto s01 = __builtin_amdgcn_permlane32_swap(we0[1], wo0[1], false, false);
                    const auto s10 = __builtin_amdgcn_permlane32_swap(we1[0], wo1[0], false, false);
                    const auto s11 = __builtin_amdgcn_permlane32_swap(we1[1], wo1[1], false, false);
                    const int c8 = ((m0 + r16) >> 3) + lhi;
                    unsigned char* o = ob + ((size_t)c8 * NPL * Tp + n + X3_HALO) * 16;
                    if (p.ablate & 32) continue;
                    *reinterpret_cast<uint4*>(o) = make_uint4(s00[0], s01[0], s00[1], s01[1]);
                    *reinterpret_cast<uint4*>(o + (size_t)Tp * 16) = make_uint4(s10[0], s11[0], s10[1], s11[1]);
                }
            }
        if (tid < 64) {                                                                // halo columns: 16 chunks x 2 planes each
            const int side = tid >> 5, c8 = (m0 >> 3) + ((tid & 31) >> 1), pl = tid & 1;
            const int tp = side == 0 ? 0 : n0 + BN + X3_HALO;
            if (side == 0 ? n0 == 0 : (n0 + BN >= nvalid && tp < Tp))
                *reinterpret_cast<uint4*>(ob + ((size_t)(c8 * NPL + pl) * Tp + tp) * 16) = make_uint4(0, 0, 0, 0);
        }
        return;
    }

    if (EPI != 2 && p.epi_vec) {
        // ---- LDS-staged epilogue (rows of y / res 16-byte aligned).  The MFMA C layout gives a lane ONE column and 16 rows: stored
        // directly that is 96 global_store_dword per lane, 256 bytes per wave-instruction, and the kernel's tail is store-ISSUE
        // bound (16 - 24 us of a 43 - 120 us launch at 240 tiles: ablation in DESIGN.md).  Instead each wave transposes its 64 x 96
        // tile through its own LDS region in four rounds of 16 rows (24 ds_write_b32, 6 ds_read_b128) and moves 16 bytes per lane:
        // 24 residual loads + 24 stores of 1 KiB per wave.  Region: 16 rows x (96 + 8) floats = 6.5 KiB per wave (the K loop's
        // stages are free once every wave has left it: one barrier).
        constexpr int EP_LD = 104;
        float* yb = p.y + (long long)b * p.y_bs;
        const float* rb = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
        const int ncol0 = n0 + wn0;
        // Residual rows: with >= 3 LDS stages (register budget 256) ALL 24 loads of the tile are in flight before the first round -
        // one memory latency for the tile instead of one per round (the rounds' LDS fences keep the compiler from hoisting them)
        constexpr bool RES_AHEAD = NSTG >= 3;
        float4 rva[RES_AHEAD ? 4 : 1][6];
        auto load_res = [&](int q, float4 (&dst)[6]) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int idx = lane + 64 * k, rl = idx / 24, c4 = idx - rl * 24, row = m0 + wm0 + (q >> 1) * 32 + (q & 1) * 16 + rl, n = ncol0 + c4 * 4;
                dst[k] = (row < p.Cout && n < nvalid) ? *reinterpret_cast<const float4*>(rb + (long long)row * p.res_cs + n)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        if (RES_AHEAD && rb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) load_res(q, rva[q]);
        }
        __syncthreads();
        float* st = reinterpret_cast<float*>(smem) + wave * (16 * EP_LD);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q >> 1, rowt0 = wm0 + i * 32 + (q & 1) * 16;          // first tile row of this round
            float4 (&rv)[6] = rva[RES_AHEAD ? q : 0];
            if (!RES_AHEAD && rb) load_res(q, rv);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = (q & 1) * 8 + rr, rl = (rr & 3) + 8 * (rr >> 2) + 4 * lhi;
                    st[rl * EP_LD + j * 32 + l31] = acc[i][j][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (EPI == 4) {
                // gated output (WN in_layers, modules.py:15-22): packed rows (2 r, 2 r + 1) -> channel r = tanh(a + g_a) * sigmoid(b + g_b),
                // g = the per-sample conditioning rows (badd).  The round's 16 rows are 8 pairs x 24 float4 columns: 3 per lane.
                const float* bad = p.badd ? p.badd + (long long)b * p.badd_bs : nullptr;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int idx = lane + 64 * k, pr = idx / 24, c4 = idx - pr * 24, rt = rowt0 + 2 * pr, row = m0 + rt, n = ncol0 + c4 * 4;
                    const float4 a4 = *reinterpret_cast<const float4*>(st + (2 * pr) * EP_LD + c4 * 4);
                    const float4 b4 = *reinterpret_cast<const float4*>(st + (2 * pr + 1) * EP_LD + c4 * 4);
                    if (row + 1 < p.Cout && n < nvalid) {
                        float ba = bias_s[rt], bb = bias_s[rt + 1];
                        if (bad) { ba += bad[row]; bb += bad[row + 1]; }
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, gv[4] = {b4.x, b4.y, b4.z, b4.w};
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ta = av[e] * XS_ACC_SCALE + ba, tb = gv[e] * XS_ACC_SCALE + bb;
                            o[e] = tanhf(ta) * sigmoidf_(tb);                       // (as conv_gemm's gate: the two paths differ by the products only)
                        }
                        float* dst = yb + (long long)(row >> 1) * p.y_cs + n;
                        if (n + 3 < nvalid) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                        else
                            for (int e = 0; e < 4 && n + e < nvalid; ++e) dst[e] = o[e];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                continue;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int idx = lane + 64 * k, rl = idx / 24, c4 = idx - rl * 24, row = m0 + rowt0 + rl, n = ncol0 + c4 * 4;
                const float4 a4 = *reinterpret_cast<const float4*>(st + rl * EP_LD + c4 * 4);
                const float bz = bias_s[rowt0 + rl];
                float o[4] = {a4.x * XS_ACC_SCALE + bz, a4.y * XS_ACC_SCALE + bz, a4.z * XS_ACC_SCALE + bz, a4.w * XS_ACC_SCALE + bz};
                if (EPI == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], p.epi_act, p.epi_slope) * p.out_scale;
                }
                if (rb) {
                    o[0] += p.res_scale * rv[k].x; o[1] += p.res_scale * rv[k].y; o[2] += p.res_scale * rv[k].z; o[3] += p.res_scale * rv[k].w;
                }
                if (row < p.Cout && n < nvalid) {
                    float* dst = yb + (long long)row * p.y_cs + n;
                    if (n + 3 < nvalid) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    else
                        for (int e = 0; e < 4 && n + e < nvalid; ++e) dst[e] = o[e];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this round's reads are done before the next round's writes
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if (EPI == 4) return;                                                // (gated: the launcher requires the staged epilogue above)
    // ---- epilogue.  C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // bias of this lane's 32 output rows from the LDS copy made at kernel start: no global latency here, no registers held
    // across the K loop
    float bv[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[i][r] = bias_s[wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
    if (EPI == 2) {
        // qkv conv of an AttentionBlock (QKVAttentionLegacy row order: head h = rows 144 h + [q 48 | k 48 | v 48]): the 4 consecutive
        // rows a lane holds per register group never straddle a section or an 8-channel chunk.  Q and K: half a chunk (4 channels
        // of one position) per lane.  V wants 4 consecutive KEYS of one channel: 4 x 4 transpose over the lane quad (DPP).
        constexpr int D = 48, KT = 64;
        const int H = p.qkv_heads, Tq = p.qkv_tq;
        const size_t qb = (size_t)2 * 6 * Tq * 16, hb = qb + (size_t)p.qkv_nt64 * (2 * (6 * KT + 384) * 16);
        unsigned char* pb = static_cast<unsigned char*>(p.qkv_planes) + (size_t)b * H * hb;
        const int q4 = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = n0 + wn0 + j * 32 + l31;
                if (((n0 + wn0 + j * 32) & ~63) >= nvalid) continue;               // its whole 64-key tile lies beyond the length (wave-uniform)
                const bool ok = n < nvalid;
                // Q and K: the 8 channels of a 16-byte chunk sit in TWO lanes (l, l + 32: rows + 4).  One v_permlane32_swap per packed
                // word pairs the register groups (2k, 2k + 1): lanes 0..31 end up with the whole even chunk, lanes 32..63 with the odd
                // one, and a lane stores 16 bytes per plane (the kernel's tail is store-issue bound: half the instructions, twice as wide).
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int r16 = m0 + wm0 + i * 32 + 16 * k2;                   // 16-row group: never straddles a head section (48 = 3 x 16)
                    if (r16 >= p.Cout) continue;
                    const int h = r16 / (3 * D), rem = r16 - h * 3 * D, sec = rem / D, c16 = rem - sec * D;
                    if (sec == 2) continue;
                    unsigned char* hp = pb + (size_t)h * hb;
                    const float sc = sec == 0 ? p.qkv_qscale * 16.f : 16.f;
                    unsigned we0[2], we1[2], wo0[2], wo1[2];
                    {
                        float ve[4], vo[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ve[e] = (acc[i][j][8 * k2 + e] * XS_ACC_SCALE + bv[i][8 * k2 + e]) * sc;
                            vo[e] = (acc[i][j][8 * k2 + 4 + e] * XS_ACC_SCALE + bv[i][8 * k2 + 4 + e]) * sc;
                            if (sec == 1 && !ok) ve[e] = vo[e] = 0.f;
                        }
                        split_pair(ve[0], ve[1], we0[0], we1[0]);
                        split_pair(ve[2], ve[3], we0[1], we1[1]);
                        split_pair(vo[0], vo[1], wo0[0], wo1[0]);
                        split_pair(vo[2], vo[3], wo0[1], wo1[1]);
                    }
                    uint4 q0, q1;
                    {
                        const auto s00 = __builtin_amdgcn_permlane32_swap(we0[0], wo0[0], false, false);
                        const auto s01 = __builtin_amdgcn_permlane32_swap(we0[1], wo0[1], false, false);
                        const auto s10 = __builtin_amdgcn_permlane32_swap(we1[0], wo1[0], false, false);
                        const auto s11 = __builtin_amdgcn_permlane32_swap(we1[1], wo1[1], false, false);
                        q0 = make_uint4(s00[0], s01[0], s00[1], s01[1]);
                        q1 = make_uint4(s10[0], s11[0], s10[1], s11[1]);
                    }
                    const int c8 = (c16 >> 3) + lhi;
                    if (sec == 0) {
                        if (ok) {
                            unsigned char* o = hp + ((size_t)c8 * Tq + n) * 16;
                            *reinterpret_cast<uint4*>(o) = q0;
                            *reinterpret_cast<uint4*>(o + (size_t)6 * Tq * 16) = q1;
                        }
                    } else {
                        unsigned char* o = hp + qb + (size_t)(n >> 6) * (2 * (6 * KT + 384) * 16) + ((size_t)c8 * KT + (n & 63)) * 16;
                        *reinterpret_cast<uint4*>(o) = q0;
                        *reinterpret_cast<uint4*>(o + 6 * KT * 16) = q1;
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int row0 = m0 + wm0 + i * 32 + 8 * rg + 4 * lhi;         // first of this lane's 4 rows
                    if (row0 >= p.Cout) continue;
                    const int h = row0 / (3 * D), rem = row0 - h * 3 * D, sec = rem / D, c = rem - sec * D;
                    if (sec != 2) continue;                                        // V only (Q / K above)
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e] * XS_ACC_SCALE + bv[i][4 * rg + e];
                    unsigned char* hp = pb + (size_t)h * hb;
                    {
                        // lane quad (keys n - q4 .. n - q4 + 3) x registers (channels c .. c + 3) -> transposed
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = ok ? v[e] * 16.f : 0.f;                 // zeros beyond the length: P = 0 there
                        {
                            const bool odd = q4 & 1;
                            const float a = odd ? t[0] : t[1], bq = odd ? t[2] : t[3];
                            const float ra = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
                            const float rb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bq), 0xB1, 0xF, 0xF, true));
                            if (odd) { t[0] = ra; t[2] = rb; } else { t[1] = ra; t[3] = rb; }
                        }
                        {
                            const bool hi2 = q4 & 2;
                            const float a = hi2 ? t[0] : t[2], bq = hi2 ? t[1] : t[3];
                            const float ra = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));
                            const float rb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bq), 0x4E, 0xF, 0xF, true));
                            if (hi2) { t[0] = ra; t[1] = rb; } else { t[2] = ra; t[3] = rb; }
                        }
                        // now: channel c + q4, keys kb .. kb + 3 with kb = n - q4 (a multiple of 4)
                        unsigned w0[2], w1[2];
                        split_pair(t[0], t[1], w0[0], w1[0]);
                        split_pair(t[2], t[3], w0[1], w1[1]);
                        // V chunk (key block u, key step j, half hh, channel): keys 32 u + 16 j + 4 hh + {0..3} | + 8 + {0..3} (attention.h)
                        const int ch = c + q4, kb = n - q4, k64 = kb & 63, u = k64 >> 5, j16 = (k64 >> 4) & 1, grp = (k64 & 15) >> 2;
                        const int chunk = ((u * 2 + j16) * 2 + (grp & 1)) * D + ch;
                        unsigned char* o = hp + qb + (size_t)(kb >> 6) * (2 * (6 * KT + 384) * 16) + (size_t)2 * 6 * KT * 16 + (size_t)chunk * 16 + (grp >> 1) * 8;
                        *reinterpret_cast<uint2*>(o) = make_uint2(w0[0], w0[1]);
                        *reinterpret_cast<uint2*>(o + 384 * 16) = make_uint2(w1[0], w1[1]);
                    }
                }
            }
        return;
    }
    float* yb = p.y + (long long)b * p.y_bs;
    const float* rb = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
    // The residual usually IS the output buffer (in-place x += f(x)): the compiler must keep every residual load behind the
    // previous store, which would serialise 64 load latencies.  Element (row, n) is read and written by this lane only, so per
    // 32 x 32 tile the 16 residual loads are issued first, into registers, and the 16 stores follow.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + wn0 + j * 32 + l31;
            if (n >= nvalid) continue;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                rv[r] = rb ? rb[(long long)(row < p.Cout ? row : p.Cout - 1) * p.res_cs + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= p.Cout) continue;
                float v = acc[i][j][r] * XS_ACC_SCALE + bv[i][r];          // undo the operands' power-of-two scales (exact)
                if (EPI == 1) v = act_apply(v, p.epi_act, p.epi_slope) * p.out_scale;
                v += p.res_scale * rv[r];
                yb[(long long)row * p.y_cs + n] = v;
            }
        }
}
}  // namespace

void launch_split_weights(const float* wp, int KW, int CinP, int CoutP, void* out, hipStream_t s) {
    DTTS_REQUIRE(CinP % 16 == 0 && CoutP % 64 == 0, "split_weights: padding");
    hipLaunchKernelGGL(split_weights_kernel, dim3(cdiv(CoutP, 256), CinP / 8, KW), dim3(256), 0, s, wp, CinP / 8, CoutP,
                       static_cast<uint4*>(out));
    DTTS_CHECK_HIP(hipGetLastError());
}

// x [B][C][T] fp32 -> the planes of the KW-tap expansion [B][KW * C/8][2][Tp][8 fp16]: chunk (tap, c8) at column t holds
// x[c][t + tap - pad] (zero outside [0, len)), so that a k = KW "same" conv over C channels is a 1x1 conv over KW * C channels whose
// weights are the same w3 image (launch_split_weights orders [tap][Cin/8]).  Columns outside [0, len) are zero chunks.
template <int KW>
__global__ __launch_bounds__(256) void split_planes_taps_kernel(const float* __restrict__ x, long long x_bs, int x_cs, const int* __restrict__ lens,
                                                               int T, int C8, int pad, int Tp, uint4* __restrict__ out, int* __restrict__ sat) {
    const int tp = blockIdx.x * 256 + threadIdx.x, c8 = blockIdx.y, b = blockIdx.z;
    if (tp >= Tp) return;
    const int t = tp - X3_HALO, len = lens ? lens[b] : T;
    const bool live = t >= 0 && t < len;
    const float* xr = x + (long long)b * x_bs + (long long)(c8 * 8) * x_cs;
    bool over = false;
#pragma unroll
    for (int tap = 0; tap < KW; ++tap) {
        uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
        const int ts = t + tap - pad;
        if (live && ts >= 0 && ts < len) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = xr[(long long)e * x_cs + ts] * XS_SCALE_X;
                over |= !(fabsf(v[e]) <= 65504.f);
            }
            split8(v, q0, q1);
        }
        uint4* o = out + ((long long)((b * KW + tap) * C8 + c8) * NPL) * Tp + tp;
        o[0] = q0;
        o[Tp] = q1;
    }
    if (sat && over) *sat = 1;
}

void launch_split_planes_taps(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int KW, int pad, void* out,
                              hipStream_t s, int* sat) {
    DTTS_REQUIRE(C % 16 == 0 && KW == 5 && pad >= 0 && pad < KW, "split_planes_taps: 5 taps, channels a multiple of 16");
    const int Tp = x3_tp(T);
    ProfScope ps("split_planes_taps_kernel", 0.0, (double)B * C * T * 4.0 * (1 + KW), s);
    hipLaunchKernelGGL((split_planes_taps_kernel<5>), dim3(cdiv(Tp, 256), C / 8, B), dim3(256), 0, s, x, x_bs, x_cs, lens, T, C / 8, pad, Tp,
                       static_cast<uint4*>(out), sat);
    DTTS_CHECK_HIP(hipGetLastError());
}

void launch_split_planes(const float* x, long long x_bs, int x_cs, const float* ab, int act, const int* lens, int T, int B, int C,
                         void* out, hipStream_t s) {
    DTTS_REQUIRE(C % 16 == 0, "split_planes: channels must be a multiple of 16");
    DTTS_REQUIRE(act == ACT_NONE || act == ACT_SILU, "split_planes: activation");
    const int Tp = x3_tp(T);
    const dim3 grid(cdiv(Tp, 256), C / 8, B);
    uint4* o = static_cast<uint4*>(out);
    const double n = (double)B * C * T;
    ProfScope ps("split_planes_kernel", 0.0, n * 8.0, s);
    if (ab) {
        if (act == ACT_SILU) hipLaunchKernelGGL((split_planes_kernel<ACT_SILU, true>), grid, dim3(256), 0, s, x, x_bs, x_cs, ab, lens, T, C / 8, Tp, o);
        else hipLaunchKernelGGL((split_planes_kernel<ACT_NONE, true>), grid, dim3(256), 0, s, x, x_bs, x_cs, ab, lens, T, C / 8, Tp, o);
    } else {
        if (act == ACT_SILU) hipLaunchKernelGGL((split_planes_kernel<ACT_SILU, false>), grid, dim3(256), 0, s, x, x_bs, x_cs, ab, lens, T, C / 8, Tp, o);
        else hipLaunchKernelGGL((split_planes_kernel<ACT_NONE, false>), grid, dim3(256), 0, s, x, x_bs, x_cs, ab, lens, T, C / 8, Tp, o);
    }
    DTTS_CHECK_HIP(hipGetLastError());
}

void launch_gn_split_planes(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int groups,
                            const float* gamma, const float* beta, float eps, const float* ada, int ada_stride, int ada_bs, int act,
                            void* out, hipStream_t s, const int* ada_idx) {
    DTTS_REQUIRE(C % groups == 0 && (C / groups) % 8 == 0 && C / groups <= 64, "gn_split_planes: group size");
    DTTS_REQUIRE(act == ACT_NONE || act == ACT_SILU, "gn_split_planes: activation");
    const int Tp = x3_tp(T);
    uint4* o = static_cast<uint4*>(out);
    ProfScope ps("gn_split_planes_kernel", 0.0, (double)B * C * T * 8.0, s);      // fp32 in, two fp16 planes out
    static const int ns = []() { const char* v = getenv("DTTS_GN_SPLIT_NS"); return v ? atoi(v) : 1; }();
    static const int nt = []() { const char* v = getenv("DTTS_GN_SPLIT_NT"); return v ? atoi(v) : 1024; }();
    static const bool reg_ok = []() { const char* v = getenv("DTTS_GN_SPLIT_REG"); return !(v && v[0] == '0'); }();
    constexpr int NI = 3;
    if (reg_ok && ns == 1 && (C / groups / 8) * Tp <= NI * 1024) {          // the slab fits the workgroup's registers: one HBM read
        if (act == ACT_SILU)
            hipLaunchKernelGGL((gn_split_planes_reg_kernel<ACT_SILU, NI>), dim3(groups, B), dim3(1024), 0, s, x, x_bs, x_cs, lens, T, C, groups,
                               gamma, beta, eps, ada, ada_stride, ada_bs, Tp, o, ada_idx);
        else
            hipLaunchKernelGGL((gn_split_planes_reg_kernel<ACT_NONE, NI>), dim3(groups, B), dim3(1024), 0, s, x, x_bs, x_cs, lens, T, C, groups,
                               gamma, beta, eps, ada, ada_stride, ada_bs, Tp, o, ada_idx);
        DTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (act == ACT_SILU)
        hipLaunchKernelGGL(gn_split_planes_kernel<ACT_SILU>, dim3(groups, B, ns), dim3(nt), 0, s, x, x_bs, x_cs, lens, T, C, groups, gamma,
                           beta, eps, ada, ada_stride, ada_bs, Tp, o, ada_idx);
    else
        hipLaunchKernelGGL(gn_split_planes_kernel<ACT_NONE>, dim3(groups, B, ns), dim3(nt), 0, s, x, x_bs, x_cs, lens, T, C, groups, gamma,
                           beta, eps, ada, ada_stride, ada_bs, Tp, o, ada_idx);
    DTTS_CHECK_HIP(hipGetLastError());
}

// Split-K scratch per launch stream: slabs of raw accumulators (96 x 256 floats per workgroup) and one arrival counter per output tile
// (zero between launches: the reducing workgroup resets it).  Launches on one stream are ordered; different streams get different slots.
namespace {
struct KSplitWs {
    float* part;
    int* count;
};
KSplitWs ksplit_workspace(hipStream_t s, size_t nslabs) {
    constexpr int SLOTS = 8;
    constexpr size_t MAX_SLABS = X3_MAX_SLABS, MAX_TILES = X3_SPLIT_COUNTERS;
    static std::mutex mu;
    static hipStream_t owner[SLOTS];
    static int owner_dev[SLOTS];
    static float* part[SLOTS];
    static int* count[SLOTS];
    static int used = 0, victim = 0;
    DTTS_REQUIRE(nslabs <= MAX_SLABS, "conv_x3 split-K: launch too large");
    int dev = 0;
    DTTS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int k = 0;
    while (k < used && !(owner[k] == s && owner_dev[k] == dev)) ++k;
    if (k == used) {
        if (used == SLOTS) {                            // every slot taken: drain the device and hand the oldest slot of this device on
            DTTS_CHECK_HIP(hipDeviceSynchronize());
            for (int t = 0; t < SLOTS; ++t, victim = (victim + 1) % SLOTS)
                if (owner_dev[victim] == dev) break;
            DTTS_REQUIRE(owner_dev[victim] == dev, "conv_x3 split-K: no scratch slot on this device");
            k = victim;
            victim = (victim + 1) % SLOTS;
            owner[k] = s;
            return {part[k], count[k]};
        }
        DTTS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&part[k]), MAX_SLABS * 96 * 256 * sizeof(float)));
        DTTS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&count[k]), MAX_TILES * sizeof(int)));
        // zeroed ON THE LAUNCH STREAM: a plain hipMemset runs on the null stream, which a non-blocking stream does not wait for - the
        // first split-K kernel of such a stream could meet recycled (non-zero) counters, or have them zeroed under it
        DTTS_CHECK_HIP(hipMemsetAsync(count[k], 0, MAX_TILES * sizeof(int), s));
        owner[k] = s;
        owner_dev[k] = dev;
        ++used;
    }
    return {part[k], count[k]};
}
}  // namespace

void x3_split_workspace(hipStream_t s, size_t nslabs, float** part, int** count) {
    const KSplitWs w = ksplit_workspace(s, nslabs);
    *part = w.part;
    *count = w.count;
}

size_t conv_x3_gn_xch_bytes(int B, int Cout, int T) { return (size_t)B * (Cout / 8) * cdiv(T, BN) * 2 * 16; }

static long long split_tiles_max() {
    static const long long v = []() { const char* e = getenv("DTTS_CONV_KSPLIT_MAXTILE"); return e ? atoll(e) : 128LL; }();
    return v;
}

bool conv_x3_gn_fusable(int Cout, int CoutP, int Cin, int KW, int groups, int B, int T) {
    if (Cout != CoutP || Cout % BM || groups <= 0 || Cout % groups || Cout / groups != 24 || Cin % 16 || (KW != 1 && KW != 3)) return false;
    const int nt = cdiv(T, BN);
    // long sequences keep the separate pass (conv_x3.h).  Split-K launches (<= 128 tiles: batches 1 - 2) carry the fused norm since
    // round 5: the tile's LAST workgroup to arrive reduces the slabs and runs the fused epilogue like any other tile (the others have
    // left by then, so the tiles a reducer may wait for are never behind more workgroups than the launch has)
    (void)B;
    return nt <= GN_FUSE_MAX_NT;
}

void launch_conv_x3(const ConvParams& p_in, hipStream_t s) {
    ConvParams p = p_in;
    DTTS_REQUIRE(p.w3 && p.x3 && (p.y || p.qkv_planes || p.gn_out3) && p.x3_tp > 0, "conv_x3: operands");
    const bool gn = p.gn_out3 != nullptr;
    static const int env_ablate = []() { const char* v = getenv("DTTS_CONV_ABLATE"); return v ? atoi(v) : 0; }();
    if (env_ablate) p.ablate = env_ablate;
    if (gn) {
        DTTS_REQUIRE(conv_x3_gn_fusable(p.Cout, p.CoutP, p.Cin, p.KW, p.gn_groups, p.B, p.Nout), "conv_x3: this launch cannot carry a fused GroupNorm");
        DTTS_REQUIRE(p.gn_gamma && p.gn_beta && p.gn_xch && p.gn_tag && p.gn_err && !p.qkv_planes && p.epi_act == ACT_NONE && p.out_scale == 1.f &&
                         (p.gn_act == ACT_NONE || p.gn_act == ACT_SILU), "conv_x3 fused GroupNorm: parameters");
        auto a16 = [](const void* q, long long bs, int cs) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0 && (bs & 3) == 0 && (cs & 3) == 0; };
        DTTS_REQUIRE((!p.y || a16(p.y, p.y_bs, p.y_cs)) && (!p.res || a16(p.res, p.res_bs, p.res_cs)), "conv_x3 fused GroupNorm: 16-byte aligned rows");
    }
    DTTS_REQUIRE(p.B > 0 && p.Nout > 0 && p.Cout > 0, "empty conv");
    DTTS_REQUIRE(p.Cin % 16 == 0 && p.CoutP % BM == 0, "conv_x3: channel padding");
    const bool gated = p.gate == GATE_TANH_SIGMOID;         // WN in_layers as a 1x1 conv over the tap-expanded planes (launch_split_planes_taps)
    DTTS_REQUIRE(p.stride == 1 && p.dil == 1 && p.phases == 1 && (p.gate == GATE_NONE ? !p.badd : gated), "conv_x3: unsupported conv form");
    DTTS_REQUIRE(!gated || (p.KW == 1 && !gn && !p.qkv_planes && !p.res && p.epi_act == ACT_NONE && p.out_scale == 1.f && p.Cout % 2 == 0),
                 "conv_x3 gated epilogue: 1x1 conv, no residual / activation");
    DTTS_REQUIRE((p.KW == 1 && p.pad == 0) || (p.KW == 3 && p.pad == 1), "conv_x3: k = 1 or k = 3 (same padding) only");
    DTTS_REQUIRE(round_up(p.Nout, BN) + 2 * X3_HALO <= p.x3_tp, "conv_x3: time padding");
    // LDS stages by launch size.  Small launches expose each workgroup's own dependency chain DMA -> barrier -> fragment reads -> MFMA,
    // so the loads run further ahead (counted vmcnt): four stages (82 KiB, one workgroup per CU) up to DTTS_CONV_STAGES4_MAXWG = 128
    // workgroups (half the CUs: batch 1), three (61 KiB, two per CU) up to DTTS_CONV_STAGES3_MAXWG = 600, two (41 KiB, three per CU)
    // for launches that fill the chip several times over.  DTTS_CONV_STAGES = 2 / 3 / 4 forces one.
    static const int force_stg = []() { const char* v = getenv("DTTS_CONV_STAGES"); const int n = v ? atoi(v) : 0; return n >= 2 && n <= 4 ? n : 0; }();
    static const long long max3 = []() { const char* v = getenv("DTTS_CONV_STAGES3_MAXWG"); return v ? atoll(v) : 600LL; }();
    static const long long max4 = []() { const char* v = getenv("DTTS_CONV_STAGES4_MAXWG"); return v ? atoll(v) : 128LL; }();
    if (gn) p.cols = nullptr;                                            // the fused GroupNorm's id order is per sample
    DTTS_REQUIRE(!p.cols || (p.ncols > 0 && p.ncols <= cdiv(p.Nout, BN) * p.B && cdiv(p.Nout, BN) < 256), "conv_x3: column table");
    const long long ntile = (long long)(p.CoutP / BM) * (p.cols ? p.ncols : cdiv(p.Nout, BN) * p.B);
    // split-K: launches of at most 128 tiles (half the CUs: batch 1) divide the channel blocks among up to 4 workgroups per tile
    static const int max_split = []() { const char* v = getenv("DTTS_CONV_KSPLIT"); const int n = v ? atoi(v) : 4; return n < 1 ? 1 : (n > 8 ? 8 : n); }();
    int S = 1;
    // (k = 3: 144 K-steps per tile; the 48 steps of a 1x1 conv barely pay for the exchange: at most 2 there)
    const long long split_tiles = split_tiles_max();
    static const long long split_wgs = []() { const char* v = getenv("DTTS_CONV_KSPLIT_WGS"); return v ? atoll(v) : 256LL; }();
    // (decided on the PADDED tile count: a column table must not change the summation order of a launch)
    const long long ntile_pad = (long long)(p.CoutP / BM) * cdiv(p.Nout, BN) * p.B;
    static const int max_split_k1 = []() { const char* v = getenv("DTTS_CONV_KSPLIT_K1"); return v ? atoi(v) : 2; }();
    if (ntile_pad <= split_tiles) S = (int)std::min<long long>(std::min<long long>(p.KW == 3 ? max_split : std::min(max_split, max_split_k1), split_wgs / ntile_pad), (p.Cin >> 4) / 8);
    if (p.ksplit_max > 0) S = std::min(S, p.ksplit_max);
    if (S < 1) S = 1;
    p.ksplit = S;
    static const bool epi_vec_on = []() { const char* v = getenv("DTTS_X3_EPI_VEC"); return !(v && v[0] == '0'); }();
    auto al16 = [](const void* q, long long bs, int cs) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0 && (bs & 3) == 0 && (cs & 3) == 0; };
    p.epi_vec = (epi_vec_on && p.y && al16(p.y, p.y_bs, p.y_cs) && (!p.res || al16(p.res, p.res_bs, p.res_cs))) ? 1 : 0;
    if (S > 1) {
        const KSplitWs w = ksplit_workspace(s, (size_t)ntile * S);
        p.kpart = w.part;
        p.kcount = w.count;
    }
    const long long nwg = ntile * S;
    static const long long max4k3 = []() { const char* v = getenv("DTTS_CONV_STAGES4_MAXWG_K3"); return v ? atoll(v) : 128LL; }();
    const int nstg = force_stg ? force_stg : (nwg <= (p.KW == 3 ? max4k3 : max4) ? 4 : (nwg <= max3 ? 3 : 2));
    const size_t lds = (size_t)nstg * (WTILE + XBUF) + BM * sizeof(float) + 16 + (gn ? 2 * BM * sizeof(float) : 0);
    const int l4 = 4 * (WTILE + XBUF) + 3 * BM * (int)sizeof(float) + 16;      // the attribute is a maximum: every instantiation gets the 4-stage size
    const dim3 grid((unsigned)nwg);
    const double cols = (double)p.B * p.Nout;
    const double flops = 2.0 * p.Cout * p.Cin * p.KW * cols;                      // fp32-equivalent; the MFMA pipe executes 3x this in fp16
    const double bytes = 4.0 * cols * p.Cin + 4.0 * cols * p.Cout * (p.res ? 2.0 : 1.0) + 4.0 * (double)p.Cout * p.Cin * p.KW;
    {
        static const bool by_shape = []() { const char* v = getenv("DTTS_PROF_SHAPES"); return v && v[0] == '1'; }();
        const char* tag = "conv_x3_kernel<128,192>";
        if (by_shape) tag = p.KW == 3 ? "conv_x3 k3" : (p.Cout > 1024 ? "conv_x3 k1 M=2304" : (p.res ? "conv_x3 k1 +res" : "conv_x3 k1"));
        ProfScope ps(tag, flops, bytes, s);
        const bool epi = p.epi_act != ACT_NONE || p.out_scale != 1.f;
#define DTTS_LAUNCH_X3(E, K3)                                                                                          \
    do {                                                                                                               \
        if (nstg == 4) { lds_optin(reinterpret_cast<const void*>(conv_x3_kernel<E, K3, 4>), l4); hipLaunchKernelGGL((conv_x3_kernel<E, K3, 4>), grid, dim3(256), lds, s, p); } \
        else if (nstg == 3) { lds_optin(reinterpret_cast<const void*>(conv_x3_kernel<E, K3, 3>), l4); hipLaunchKernelGGL((conv_x3_kernel<E, K3, 3>), grid, dim3(256), lds, s, p); } \
        else { lds_optin(reinterpret_cast<const void*>(conv_x3_kernel<E, K3, 2>), l4); hipLaunchKernelGGL((conv_x3_kernel<E, K3, 2>), grid, dim3(256), lds, s, p); } \
    } while (0)
        if (gn) {
            if (p.KW == 3) DTTS_LAUNCH_X3(3, true);
            else DTTS_LAUNCH_X3(3, false);
        } else if (gated) {
            DTTS_REQUIRE(p.epi_vec, "conv_x3 gated epilogue: rows of y must be 16-byte aligned");
            DTTS_LAUNCH_X3(4, false);
        } else if (p.qkv_planes) {
            DTTS_REQUIRE(p.KW == 1 && !epi && !p.res && p.Cout % 144 == 0 && p.qkv_heads * 144 == p.Cout, "qkv planes epilogue: 48-channel heads, 1x1 conv");
            DTTS_LAUNCH_X3(2, false);
        } else if (p.KW == 3) {
            if (epi) DTTS_LAUNCH_X3(1, true);
            else DTTS_LAUNCH_X3(0, true);
        } else {
            if (epi) DTTS_LAUNCH_X3(1, false);
            else DTTS_LAUNCH_X3(0, false);
        }
#undef DTTS_LAUNCH_X3
    }
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
