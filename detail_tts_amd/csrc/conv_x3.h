// fp32-accurate Conv1d on the bf16 matrix pipe: every fp32 operand is split into three bf16 planes (a = a0 + a1 + a2, exact to
// 24 bits) and the six significant cross products a_i * b_j (i + j <= 2) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// Measured error vs an fp64 reference is the same as the fp32 MFMA path's (rel. 7e-7, tools/ubench/gemm_x3.hip) at 1.7x its
// throughput (6 bf16 MFMAs per 16 channels cost 192 SIMD-cycles against 512 for 8 v_mfma_f32_32x32x2_f32).
//
// Operand layout in HBM — 16-byte chunks = 8 consecutive channels of one plane:
//   w3 [tap][Cin/8][3][CoutP][8 bf16]    split once at bind time from the fp32 packed weights  (launch_split_weights)
//   x3 [b][Cin/8][3][Tp][8 bf16]         written by the producer side (launch_split_planes: GroupNorm affine + SiLU + zero
//                                        padding + per-sample length folded in), Tp = round_up(T,128) + 2 halo columns
// A K-step (16 channels) of a 128-row tile is 6 contiguous 2 KiB runs in HBM and in LDS, so both tiles are moved by LDS-DMA and
// every ds_read_b128 of an MFMA fragment is bank-conflict-free with no padding or swizzle.
#pragma once
#include "conv_gemm.h"

namespace dtts {

constexpr int X3_HALO = 1;   // zero columns on each side of the time axis (covers k = 3, dilation 1)
static inline int x3_tp(int T) { return round_up(T, 128) + 2 * X3_HALO; }
static inline size_t x3_bytes(int B, int C, int T) { return (size_t)B * (C / 8) * 3 * x3_tp(T) * 16; }

// wp: fp32 packed weights [KW][CinP][CoutP] -> out [KW][CinP/8][3][CoutP][8 bf16]
void launch_split_weights(const float* wp, int KW, int CinP, int CoutP, void* out, hipStream_t s);
// x [B][C][T] fp32 (strides) -> x3; v = act(a*x + d) with (a, d) = ab[b][c][0..1] (ab may be null); zero outside [0, len[b])
void launch_split_planes(const float* x, long long x_bs, int x_cs, const float* ab, int act, const int* lens, int T, int B, int C,
                         void* out, hipStream_t s);
// GroupNorm (statistics + affine, optional AdaGN (1 + scale, shift) from `ada`) + activation + split in one pass; arguments as
// launch_gn_coeffs (ops.h)
void launch_gn_split_planes(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int groups,
                            const float* gamma, const float* beta, float eps, const float* ada, int ada_stride, int ada_bs, int act,
                            void* out, hipStream_t s, const int* ada_idx = nullptr);
// uses p.w3 / p.x3 / p.x3_tp (+ the epilogue fields of ConvParams); stride 1, dilation 1, pad <= X3_HALO, no gate / phases / badd
void launch_conv_x3(const ConvParams& p, hipStream_t s);

}  // namespace dtts
