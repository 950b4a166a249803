// fp32-accurate Conv1d on the 16-bit matrix pipe ("split precision"): every fp32 operand, scaled by a power of two, is split into
// TWO fp16 planes (a s = h0 + h1, 22 significant bits; both differences exact in fp32) and the three significant cross products
// h0 h0', h0 h1', h1 h0' are accumulated in fp32 by v_mfma_f32_32x32x16_f16 (the dropped h1 h1' term is < 2^-22 relative).
// Measured against an fp64 reference the result carries the error of an fp32 GEMM - rel. 4.5e-7 at K = 768, below a 3 x bf16 / 6
// product split (6.8e-7: twice the accumulations) and below numpy's sgemm (7.0e-7) - tools/ubench/gemm_x3v.hip - at 3 matrix
// instructions per 16 channels instead of 8 v_mfma_f32_32x32x2_f32 (96 SIMD-cycles against 512).
//
// Operand layout in HBM - 16-byte chunks = 8 consecutive channels of one plane:
//   w3 [tap][Cin/8][2][CoutP][8 fp16]    split once at bind time from the fp32 packed weights  (launch_split_weights), x XS_SCALE_W
//   x3 [b][Cin/8][2][Tp][8 fp16]         written by the producer side (launch_split_planes / launch_gn_split_planes: GroupNorm affine
//                                        + SiLU + zero padding + per-sample length folded in), x XS_SCALE_X; Tp = round_up(T, 192) + 2
//                                        halo columns
// A K-step (16 channels) of a tile is 4 contiguous runs (plane, k-half) in HBM and in LDS, so both tiles are moved by LDS-DMA and every
// ds_read_b128 of an MFMA fragment is bank-conflict-free with no padding or swizzle.
#pragma once
#include "conv_gemm.h"

namespace dtts {

constexpr int X3_HALO = 1;   // zero columns on each side of the time axis (covers k = 3, dilation 1)
constexpr int X3_BN = 192;   // N tile of the conv kernel: 936 mel frames = 4.875 tiles (2.5 % padding; 128 would pad 8.6 %)
static inline int x3_tp(int T) { return round_up(T, X3_BN) + 2 * X3_HALO; }
static inline size_t x3_bytes(int B, int C, int T) { return (size_t)B * (C / 8) * 2 * x3_tp(T) * 16; }

// wp: fp32 packed weights [KW][CinP][CoutP] -> out [KW][CinP/8][2][CoutP][8 fp16]
void launch_split_weights(const float* wp, int KW, int CinP, int CoutP, void* out, hipStream_t s);
// x [B][C][T] fp32 (strides) -> x3; v = act(a*x + d) with (a, d) = ab[b][c][0..1] (ab may be null); zero outside [0, len[b])
void launch_split_planes(const float* x, long long x_bs, int x_cs, const float* ab, int act, const int* lens, int T, int B, int C,
                         void* out, hipStream_t s);
// the KW-tap expansion of x as planes [B][KW * C/8][2][x3_tp(T)][8 fp16] (chunk (tap, c8), column t = x[c][t + tap - pad], zero outside
// [0, len)): a k = KW conv with "same" padding becomes a 1x1 conv_x3 launch over KW * C input channels on the SAME w3 image - the route
// the flow's WaveNet in_layers (k = 5, gated) take onto conv_x3's small-launch pipeline.  sat as launch_split_planes_ex.
void launch_split_planes_taps(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int KW, int pad, void* out,
                              hipStream_t s, int* sat = nullptr);
// GroupNorm (statistics + affine, optional AdaGN (1 + scale, shift) from `ada`) + activation + split in one pass; arguments as
// launch_gn_coeffs (ops.h)
void launch_gn_split_planes(const float* x, long long x_bs, int x_cs, const int* lens, int T, int B, int C, int groups,
                            const float* gamma, const float* beta, float eps, const float* ada, int ada_stride, int ada_bs, int act,
                            void* out, hipStream_t s, const int* ada_idx = nullptr);
// uses p.w3 / p.x3 / p.x3_tp (+ the epilogue fields of ConvParams); stride 1, dilation 1, pad <= X3_HALO, no phases; gate (tanh * sigmoid on packed row pairs) + badd only
// as a 1x1 conv without residual (EPI 4: the WaveNet in_layers over launch_split_planes_taps planes)
void launch_conv_x3(const ConvParams& p, hipStream_t s);

// Split scratch of a launch stream (conv_x3's split-K slabs; the trunk attention's key-split partials use the same slot: launches on one
// stream are ordered): `part` = nslabs slabs of X3_SLAB_FLOATS floats, `count` = X3_SPLIT_COUNTERS arrival counters that are zero
// between launches (the reducing workgroup / wave resets its counter).
constexpr size_t X3_SLAB_FLOATS = 96 * 256, X3_MAX_SLABS = 1024, X3_SPLIT_COUNTERS = 4096;
void x3_split_workspace(hipStream_t s, size_t nslabs, float** part, int** count);

// ---- fused GroupNorm (p.gn_out3): the norm + activation + split that FOLLOWS a trunk conv runs in that conv's epilogue.  A tile holds
// 128 rows x 192 columns of one sample; a GroupNorm group is 24 channels x all T columns, so a tile needs the statistics of the <= 7
// groups its rows touch over ALL N tiles (and, for the groups that straddle its row range, of the neighbouring M tile).  Every wave
// publishes {mean, M2, count, tag} of its 8 row chunks x 96 columns as 16-byte words (agent scope, the token kernel's protocol), a
// tile polls the words of its groups, combines them in a fixed order (Chan's parallel variance: deterministic, no cancellation) and
// normalises its accumulators where they sit.  Tiles of a fused launch are ordered (sample, M tile, N tile): a tile waits only for
// tiles at most 2 N - 1 positions ahead of it in dispatch order, so at most that many workgroups per XCD can ever be waiting for an
// undispatched one - with N <= GN_FUSE_MAX_NT that is far below the workgroup slots of an XCD even with a second fused launch on
// another stream and the persistent GPT token kernel holding CUs.  Longer sequences keep the separate gn_split_planes
// pass.  Split-K launches (batches 1 - 2) carry it too (round 5): only a tile's reducing workgroup reaches the epilogue - built for
// configs[1] (a batch-1 forward is a chain of ~125 launches of 10 - 40 us), measured SLOWER there as well (forward pair 2751 -> 2896 us:
// the fused tails cost a conv ~10 us, the pass they replace ~5 us + a launch gap), so it stays an option (gn_fuse), default off.
constexpr int GN_FUSE_MAX_NT = 6;
size_t conv_x3_gn_xch_bytes(int B, int Cout, int T);
// whether launch_conv_x3 can run p with the fused norm (else: conv to p.y, then launch_gn_split_planes)
bool conv_x3_gn_fusable(int Cout, int CoutP, int Cin, int KW, int groups, int B, int T);

// ---- dilated / wide-kernel variant (conv_x3d.hip): HiFiGAN ResBlock1 convs, k = 3 / 7 / 11, any dilation with (k - 1) dil <= 64
constexpr int X3D_HALO = 32;                          // left zero columns of its planes (>= the largest pad: (11 * 5 - 5) / 2 = 25)
static inline int x3d_tp(int T) { return round_up(T, X3_BN) + 64 + X3D_HALO; }
static inline size_t x3d_bytes(int B, int CP, int T) { return (size_t)B * (CP / 8) * 2 * x3d_tp(T) * 16; }
// x [B][C][T] fp32 -> planes [B][CP/8][2][Tp][8 fp16], `halo` zero columns on the left, zero chunks for channels C .. CP - 1
// sat (may be null): device flag set to 1 when a scaled value lies beyond fp16's range (the planes saturate there)
void launch_split_planes_ex(const float* x, long long x_bs, int x_cs, int act, float slope, const int* lens, int T, int B, int C, int CP,
                            int halo, int Tp, void* out, hipStream_t s, int* sat = nullptr);
// p.w3 / p.x3 / p.x3_tp / p.x3_halo, p.Cin = padded input channels; stride 1, no gate / phases / badd.  p.next3: act(y) written also
// (p.y == null: only) as the planes of the NEXT conv - live columns; the buffer's margins are zeroed by launch_zero_plane_margins
void launch_conv_x3d(const ConvParams& p, hipStream_t s);
void launch_zero_plane_margins(const int* lens, int T, int B, int CP, int halo, int Tp, void* out, hipStream_t s);

}  // namespace dtts
