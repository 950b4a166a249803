// Split-precision Conv1d with kernel width 3 / 7 / 11 and dilation (HiFiGAN ResBlock1 of the wide generator stages,
// vqvae/modules/modules.py:240-334): same arithmetic, operand layouts and 128 x 192 tile as conv_x3.hip (two scaled fp16 planes per
// operand, three v_mfma_f32_32x32x16_f16 products per fp32 product), generalised along the taps:
//   * the X tile of a 16-channel block carries its dilated halo - 192 + (KW - 1) dil <= 242 columns, staged as 256 columns per
//     (plane, k-half) "kind" - and is fetched ONCE per block; tap t reads it through a ds_read_b128 shifted by t dil columns;
//   * channels need not be a multiple of 16 / 128: the packed weights are zero-padded (CinP, CoutP) and the producer writes zero chunks
//     for the padded input channels (launch_split_planes_ex);
//   * the time axis of the planes carries `x3_halo` zero columns on the left (>= pad) and enough on the right for the last tile.
// These launches fill the chip several times over (thousands of tiles), so two LDS stages with a plain vmcnt(0) per K-step suffice.
#include <cstdlib>

#include "conv_x3.h"
#include "prof.h"
#include "split3.h"

namespace dtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int NPL = XS_PLANES, NK = 2 * NPL;
constexpr int BN = X3_BN, XCOLS = 256;
constexpr int XTILE = NK * XCOLS * 16;               // 16 KiB

// MI: 32-row MFMA blocks per wave: M tile 128 (MI = 2) or 64 (MI = 1: 50-channel stage, CoutP = 64)
// NEXT: the epilogue writes the next conv's planes (ConvParams::next3) - its own instantiation: with both epilogues in one kernel the
// 128-row form no longer fits the 168 registers three workgroups per CU leave (350 spills, 2.7 x slower)
template <int KW, int MI, bool NEXT>
__global__ __launch_bounds__(256, (NEXT && MI == 2 && KW > 3) ? 2 : 3) void conv_x3d_kernel(ConvParams p) {
    constexpr int BM = 64 * MI, WTILE = NK * BM * 16, XOFF = 2 * WTILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int mtiles = p.CoutP / BM, ntiles = (p.Nout + BN - 1) / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % mtiles, nb = L / mtiles;
    const int b = nb / ntiles;
    const int m0 = mt * BM, n0 = (nb - b * ntiles) * BN;
    const int nvalid = p.len_out ? p.len_out[b] : p.Nout;
    if (n0 >= nvalid) return;
    const int C8 = p.Cin >> 3, c16n = p.Cin >> 4, Tp = p.x3_tp, dil = p.dil;
    const uint4* wbase = static_cast<const uint4*>(p.w3) + m0 + lane;
    // buffer column j of the X tile = plane column n0 + j + (halo - pad): output n, tap t reads buffer column (n - n0) + t dil
    const uint4* xbase = static_cast<const uint4*>(p.x3) + (long long)b * C8 * NPL * Tp + n0 + (p.x3_halo - p.pad) + lane;
    const long long wtap = (long long)C8 * NPL * p.CoutP;

    auto dma = [&](const uint4* g, int lds_off) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(smem + lds_off), 16, 0, 0);
    };
    auto w_piece = [&](int j, int c16, int tap, int stage) {            // j = kind * MI + 64-row block
        const int kind = j / MI, pl = kind >> 1, h = kind & 1, rh = j - kind * MI;
        dma(wbase + tap * wtap + ((long long)(2 * c16 + h) * NPL + pl) * p.CoutP + rh * 64, stage * WTILE + kind * (BM * 16) + rh * 1024);
    };
    auto x_piece = [&](int j, int c16, int stage) {                      // j = kind * 4 + column block
        const int kind = j >> 2, pl = kind >> 1, h = kind & 1, cb = j & 3;
        dma(xbase + ((long long)(2 * c16 + h) * NPL + pl) * Tp + cb * 64, XOFF + stage * XTILE + kind * (XCOLS * 16) + cb * 1024);
    };

    const int wm0 = (wave >> 1) * (32 * MI), wn0 = (wave & 1) * 96;
    f32x16 acc[MI][3];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: W of step (block 0, tap 0) and X of block 0
#pragma unroll
    for (int i = 0; i < MI; ++i) w_piece(wave * MI + i, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) x_piece(wave + 4 * k, 0, 0);
    float* bias_s = reinterpret_cast<float*>(smem + XOFF + 2 * XTILE);
    if (tid < BM) bias_s[tid] = (p.bias && m0 + tid < p.Cout) ? p.bias[m0 + tid] : 0.f;

    for (int c16 = 0; c16 < c16n; ++c16) {
        const unsigned char* Xb = smem + XOFF + (c16 & 1) * XTILE;
        const bool more_blocks = c16 + 1 < c16n;
#pragma unroll
        for (int tap = 0; tap < KW; ++tap) {
            const int ks = c16 * KW + tap;                               // KW is odd: the W stage alternates with the step parity
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // everything issued during the previous step has landed
            __builtin_amdgcn_s_barrier();                                // ... for every wave; and the stages refilled below are free
            const unsigned char* As = smem + (ks & 1) * WTILE + lhi * (BM * 16);
            hf8 a[MI][NPL], bb[3][NPL];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned char* xq = Xb + lhi * (XCOLS * 16) + (wn0 + j * 32 + l31 + tap * dil) * 16;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bb[j][pl] = *reinterpret_cast<const hf8*>(xq + pl * (2 * XCOLS * 16));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) a[i][pl] = *reinterpret_cast<const hf8*>(As + pl * (2 * BM * 16) + (wm0 + i * 32 + l31) * 16);
            const bool last_tap = tap == KW - 1;
            const bool next_w = !last_tap || more_blocks;
            const int nc16 = last_tap ? c16 + 1 : c16, ntap = last_tap ? 0 : tap + 1;
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
            int slot = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][TA[t]], bb[j][TB[t]], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // one LDS-DMA issue point after every three MFMAs (3 MI slots per step): the wave's MI W pieces of the next step first,
                    // then its 4 X pieces of the next block, piece k at tap k KW / 4
                    if (slot < MI) {
                        if (next_w) w_piece(wave * MI + slot, nc16, ntap, (ks + 1) & 1);
                    }
                    if (more_blocks) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int xslot = MI == 2 ? 2 + k : ((KW == 3 && k == 1) ? 2 : 1);
                            if (tap == k * KW / 4 && slot == xslot) x_piece(wave + 4 * k, c16 + 1, (c16 + 1) & 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    ++slot;
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (NEXT) {
        // The consumer's operand planes straight from the accumulators (as conv_x3.hip's fused-GroupNorm tail): a lane holds 4 + 4 rows of
        // two 8-channel chunks of ONE column; one v_permlane32_swap per packed word pairs the lane halves so that lanes 0..31 own the even
        // chunk, lanes 32..63 the odd one, 16 bytes per plane.  Columns of the tile beyond the length are written as zeros.  Eight
        // accumulators at a time, bias from its LDS copy: the kernel's register budget (168: three workgroups per CU) is the K loop's.
        float* ybn = p.y ? p.y + (long long)b * p.y_bs : nullptr;
        const float* rbn = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
        const bool epin = p.epi_act != ACT_NONE || p.out_scale != 1.f;
        unsigned char* ob = static_cast<unsigned char*>(p.next3) + (size_t)b * p.next_c8 * NPL * p.next_tp * 16;
        bool over = false;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = n0 + wn0 + j * 32 + l31;
                const bool ok = n < nvalid;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int r16 = wm0 + i * 32 + 16 * k2;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = m0 + r16 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                        v[e] = (rbn && ok) ? rbn[(long long)(row < p.Cout ? row : p.Cout - 1) * p.res_cs + n] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int rt = r16 + (e & 3) + 8 * (e >> 2) + 4 * lhi, row = m0 + rt;
                        float u = acc[i][j][8 * k2 + e] * XS_ACC_SCALE + bias_s[rt];
                        if (epin) u = act_apply(u, p.epi_act, p.epi_slope) * p.out_scale;
                        u += p.res_scale * v[e];
                        const bool live = ok && row < p.Cout;
                        if (live && ybn) ybn[(long long)row * p.y_cs + n] = u;
                        if (p.next_act == ACT_LRELU) u = u > 0.f ? u : u * p.next_slope;
                        u *= XS_SCALE_X;
                        over |= live && !(fabsf(u) <= 65504.f);
                        v[e] = live ? u : 0.f;
                    }
                    unsigned we0[2], we1[2], wo0[2], wo1[2];
                    split_pair(v[0], v[1], we0[0], we1[0]);
                    split_pair(v[2], v[3], we0[1], we1[1]);
                    split_pair(v[4], v[5], wo0[0], wo1[0]);
                    split_pair(v[6], v[7], wo0[1], wo1[1]);
                    const auto s00 = __builtin_amdgcn_permlane32_swap(we0[0], wo0[0], false, false);
                    const auto s01 = __builtin_amdgcn_permlane32_swap(we0[1], wo0[1], false, false);
                    const auto s10 = __builtin_amdgcn_permlane32_swap(we1[0], wo1[0], false, false);
                    const auto s11 = __builtin_amdgcn_permlane32_swap(we1[1], wo1[1], false, false);
                    const int c8 = ((m0 + r16) >> 3) + lhi;
                    if (c8 < p.next_c8) {
                        unsigned char* o = ob + ((size_t)c8 * NPL * p.next_tp + n + p.next_halo) * 16;
                        *reinterpret_cast<uint4*>(o) = make_uint4(s00[0], s01[0], s00[1], s01[1]);
                        *reinterpret_cast<uint4*>(o + (size_t)p.next_tp * 16) = make_uint4(s10[0], s11[0], s10[1], s11[1]);
                    }
                }
            }
        if (p.next_sat && over) *p.next_sat = 1;
        return;
    }
    // ---- epilogue (as conv_x3.hip, EPI 0 / 1): bias, activation / out_scale, residual, var-len masking
    float bv[MI][16];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[i][r] = bias_s[wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
    float* yb = p.y + (long long)b * p.y_bs;
    const float* rb = p.res ? p.res + (long long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs : nullptr;
    const bool epi = p.epi_act != ACT_NONE || p.out_scale != 1.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
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
                float v = acc[i][j][r] * XS_ACC_SCALE + bv[i][r];
                if (epi) v = act_apply(v, p.epi_act, p.epi_slope) * p.out_scale;
                v += p.res_scale * rv[r];
                yb[(long long)row * p.y_cs + n] = v;
            }
        }
}

// fp32 [B][C][T] -> planes [B][CP/8][2][Tp][8 fp16] with `halo` zero columns on the left; v = act(x) (ACT_NONE / ACT_LRELU(slope) /
// ACT_SILU); channels >= C and columns outside [0, len) are zero
template <int ACT>
__global__ __launch_bounds__(256) void split_planes_ex_kernel(const float* __restrict__ x, long long x_bs, int x_cs, float slope,
                                                             const int* __restrict__ lens, int T, int C, int C8P, int halo, int Tp,
                                                             uint4* __restrict__ out, int* __restrict__ sat) {
    const int tp = blockIdx.x * 256 + threadIdx.x, c8 = blockIdx.y, b = blockIdx.z;
    if (tp >= Tp) return;
    const int t = tp - halo, len = lens ? lens[b] : T;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (t >= 0 && t < len && c8 * 8 < C) {
        const float* xr = x + (long long)b * x_bs + (long long)(c8 * 8) * x_cs + t;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float u = (c8 * 8 + e < C) ? xr[(long long)e * x_cs] : 0.f;
            if (ACT == ACT_LRELU) u = u > 0.f ? u : u * slope;
            if (ACT == ACT_SILU) u = u * __frcp_rn(1.f + __expf(-u));
            v[e] = u * XS_SCALE_X;
        }
        if (sat) {               // range check (option x3_range_check): split8 clamps beyond +-65504, silently wrong from there on
            float m = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
            if (!(m <= 65504.f)) *sat = 1;
        }
        split8(v, q0, q1);
    }
    uint4* o = out + ((long long)(b * C8P + c8) * NPL) * Tp + tp;
    o[0] = q0;
    o[Tp] = q1;
}
}  // namespace

void launch_split_planes_ex(const float* x, long long x_bs, int x_cs, int act, float slope, const int* lens, int T, int B, int C, int CP,
                            int halo, int Tp, void* out, hipStream_t s, int* sat) {
    DTTS_REQUIRE(CP % 16 == 0 && CP >= C && halo >= 0 && Tp >= T + halo, "split_planes_ex: padding");
    DTTS_REQUIRE(act == ACT_NONE || act == ACT_LRELU || act == ACT_SILU, "split_planes_ex: activation");
    const dim3 grid(cdiv(Tp, 256), CP / 8, B);
    uint4* o = static_cast<uint4*>(out);
    ProfScope ps("split_planes_kernel", 0.0, (double)B * C * T * 8.0, s);
    if (act == ACT_LRELU) hipLaunchKernelGGL(split_planes_ex_kernel<ACT_LRELU>, grid, dim3(256), 0, s, x, x_bs, x_cs, slope, lens, T, C, CP / 8, halo, Tp, o, sat);
    else if (act == ACT_SILU) hipLaunchKernelGGL(split_planes_ex_kernel<ACT_SILU>, grid, dim3(256), 0, s, x, x_bs, x_cs, slope, lens, T, C, CP / 8, halo, Tp, o, sat);
    else hipLaunchKernelGGL(split_planes_ex_kernel<ACT_NONE>, grid, dim3(256), 0, s, x, x_bs, x_cs, slope, lens, T, C, CP / 8, halo, Tp, o, sat);
    DTTS_CHECK_HIP(hipGetLastError());
}

namespace {
// zero margins of a planes buffer that a conv epilogue fills (ConvParams::next3): columns [0, halo) and [halo + len, Tp) of every chunk row
__global__ __launch_bounds__(256) void zero_plane_margins_kernel(const int* __restrict__ lens, int T, int C8P, int halo, int Tp, uint4* __restrict__ out) {
    const int row = blockIdx.x, b = blockIdx.y;                    // row = chunk * NPL + plane
    const int len = lens ? lens[b] : T, right = halo + len, nz = halo + (Tp - right);
    uint4* o = out + ((long long)b * C8P * NPL + row) * Tp;
    for (int i = threadIdx.x; i < nz; i += 256) o[i < halo ? i : right + (i - halo)] = make_uint4(0, 0, 0, 0);
}
}  // namespace

void launch_zero_plane_margins(const int* lens, int T, int B, int CP, int halo, int Tp, void* out, hipStream_t s) {
    DTTS_REQUIRE(CP % 8 == 0 && halo >= 0 && Tp >= T + halo, "zero_plane_margins: padding");
    hipLaunchKernelGGL(zero_plane_margins_kernel, dim3(CP / 8 * NPL, B), dim3(256), 0, s, lens, T, CP / 8, halo, Tp, static_cast<uint4*>(out));
    DTTS_CHECK_HIP(hipGetLastError());
}

// p.w3 / p.x3 / p.x3_tp / p.x3_halo; p.Cin = the PADDED input channels (multiple of 16); stride 1, no gate / phases / badd
void launch_conv_x3d(const ConvParams& p, hipStream_t s) {
    DTTS_REQUIRE(p.w3 && p.x3 && (p.y || p.next3) && p.x3_tp > 0, "conv_x3d: operands");
    DTTS_REQUIRE(!p.next3 || (p.next_c8 > 0 && p.next_halo >= 0 && round_up(p.Nout, BN) + p.next_halo <= p.next_tp &&
                              (p.next_act == ACT_NONE || p.next_act == ACT_LRELU)), "conv_x3d: planes for the next conv");
    DTTS_REQUIRE(p.B > 0 && p.Nout > 0 && p.Cout > 0, "empty conv");
    DTTS_REQUIRE(p.Cin % 16 == 0 && p.CoutP % 64 == 0, "conv_x3d: channel padding");
    DTTS_REQUIRE(p.stride == 1 && p.phases == 1 && p.gate == GATE_NONE && !p.badd && p.dil >= 1, "conv_x3d: unsupported conv form");
    DTTS_REQUIRE(p.KW == 3 || p.KW == 7 || p.KW == 11, "conv_x3d: kernel width 3, 7 or 11");
    DTTS_REQUIRE(BN + (p.KW - 1) * p.dil <= XCOLS && p.pad <= p.x3_halo, "conv_x3d: dilated halo exceeds the staged tile");
    DTTS_REQUIRE(round_up(p.Nout, BN) + (XCOLS - BN) + p.x3_halo <= p.x3_tp, "conv_x3d: time padding");
    const int MI = p.CoutP % 128 == 0 ? 2 : 1, BM = 64 * MI;
    const size_t lds = (size_t)2 * NK * BM * 16 + 2 * XTILE + BM * sizeof(float);
    const int lmax = 2 * NK * 128 * 16 + 2 * XTILE + 128 * (int)sizeof(float);
    const dim3 grid((unsigned)((long long)(p.CoutP / BM) * cdiv(p.Nout, BN) * p.B));
    const double cols = (double)p.B * p.Nout;
    ProfScope ps(MI == 2 ? "conv_x3d_kernel<128,192>" : "conv_x3d_kernel<64,192>", 2.0 * p.Cout * p.Cin * p.KW * cols,
                 4.0 * cols * p.Cin + 4.0 * cols * p.Cout * (p.res ? 2.0 : 1.0) + 4.0 * (double)p.Cout * p.Cin * p.KW, s);
#define DTTS_LAUNCH_X3D(K)                                                                            \
    do {                                                                                              \
        if (MI == 2 && p.next3) { lds_optin(reinterpret_cast<const void*>(conv_x3d_kernel<K, 2, true>), lmax); hipLaunchKernelGGL((conv_x3d_kernel<K, 2, true>), grid, dim3(256), lds, s, p); } \
        else if (MI == 2) { lds_optin(reinterpret_cast<const void*>(conv_x3d_kernel<K, 2, false>), lmax); hipLaunchKernelGGL((conv_x3d_kernel<K, 2, false>), grid, dim3(256), lds, s, p); } \
        else if (p.next3) { lds_optin(reinterpret_cast<const void*>(conv_x3d_kernel<K, 1, true>), lmax); hipLaunchKernelGGL((conv_x3d_kernel<K, 1, true>), grid, dim3(256), lds, s, p); } \
        else { lds_optin(reinterpret_cast<const void*>(conv_x3d_kernel<K, 1, false>), lmax); hipLaunchKernelGGL((conv_x3d_kernel<K, 1, false>), grid, dim3(256), lds, s, p); } \
    } while (0)
    if (p.KW == 3) DTTS_LAUNCH_X3D(3);
    else if (p.KW == 7) DTTS_LAUNCH_X3D(7);
    else DTTS_LAUNCH_X3D(11);
#undef DTTS_LAUNCH_X3D
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
