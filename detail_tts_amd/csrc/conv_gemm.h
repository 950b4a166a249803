// Conv1d / Linear / ConvTranspose1d as an fp32-MFMA GEMM over [B, C, T] activations.
//
//   y[b, co, n] = epi( sum_{tap, ci} Wp[tap][ci][co] * pro(x[b, ci, n*stride + tap*dil - pad]) )
//
// Weights are pre-packed by the host (detail_tts_amd/packing.py) K-major:
//   Wp[(tap * CinP + ci) * CoutP + co], CinP % 16 == 0, CoutP % BM == 0, zero padded,
// so the kernel never bounds-checks a weight load.
#pragma once
#include "common.h"

namespace dtts {

struct ConvParams {
    // input
    const float* x = nullptr;
    long long x_bs = 0;      // batch stride (floats)
    int x_cs = 0;            // channel stride (floats) == allocated T of the input buffer
    const int* len_in = nullptr;   // [B] valid input length per sample (null -> Tin)
    const int* x_bidx = nullptr;   // optional [B]: input sample index of output sample b (shared inputs across the batch)
    int Tin = 0;
    // prologue: v = act(a*x + d) with (a,d) = pro_ab[b][ci][0..1]; positions outside [0,len) are 0
    const float* pro_ab = nullptr;
    int pro_act = ACT_NONE;
    float pro_slope = 0.f;
    // weights
    const float* w = nullptr;
    const float* bias = nullptr;   // [CoutP] in packed row order (or null)
    int Cin = 0, CinP = 0, Cout = 0, CoutP = 0, KW = 1, stride = 1, dil = 1, pad = 0;
    // per-sample per-row additive term (packed row order), e.g. WN cond_layer slice or Generator.cond(g)
    const float* badd = nullptr;
    int badd_bs = 0;
    // epilogue
    int gate = GATE_NONE;          // pairs packed rows (2r, 2r+1) -> output channel r
    int epi_act = ACT_NONE;
    float epi_slope = 0.f;
    float out_scale = 1.f;         // applied after act, before residual
    const float* res = nullptr;    // residual, same indexing as y
    long long res_bs = 0;
    int res_cs = 0;
    float res_scale = 1.f;
    int res_bmod = 0;              // residual batch index = b % res_bmod (0: b) — shared residual across batch halves
    int mask_out = 0;              // unused rows/cols are never written; kept for clarity
    // output
    float* y = nullptr;
    long long y_bs = 0;
    int y_cs = 0;
    const int* len_out = nullptr;  // [B] valid number of output columns n per sample (null -> Nout)
    int Nout = 0;                  // number of GEMM columns per sample (max)
    int phases = 1;                // ConvTranspose: packed row = ph*Cout + co, t_out = n*phases + ph
    int B = 0;
    // split-precision path (conv_x3.h): pre-split operands; when both are set launch_conv_x3 is used instead
    const void* w3 = nullptr;
    const void* x3 = nullptr;
    int x3_tp = 0;
    int x3_halo = 1;               // zero columns left of t = 0 in the planes (conv_x3d: >= pad; conv_x3 uses X3_HALO = 1)
    // conv_x3 only, the trunk's qkv conv: write the output as the split-precision attention's operand images (attention.h:
    // AttnPlanes) instead of fp32 rows - Q / K chunks straight from the accumulator layout, V through a 4 x 4 lane transpose
    void* qkv_planes = nullptr;
    int qkv_heads = 0, qkv_nt64 = 0, qkv_tq = 0;
    float qkv_qscale = 1.f;        // softmax scale * log2(e), folded into the Q planes
    // conv_x3 only, filled by its launcher: split-K for launches far smaller than the chip (batch 1).  The channel blocks are divided
    // among `ksplit` workgroups per output tile; each leaves its raw accumulators in `kpart`, the last to arrive (counter in `kcount`)
    // sums them in split order - deterministic - and runs the epilogue.
    // conv_x3 only: the GroupNorm that FOLLOWS this conv folded into its epilogue (conv_x3.h "fused GroupNorm"): the tile publishes
    // partial statistics of its output, waits for the partials of the groups its rows belong to, applies GN (x AdaGN (1 + scale) + shift),
    // the activation and the fp16 split in registers and writes the consumer's operand planes.  y may then be null (no fp32 output).
    void* gn_out3 = nullptr;           // planes of the normalised output [B][Cout/8][2][x3_tp][8 fp16]; null = no fused norm
    const float *gn_gamma = nullptr, *gn_beta = nullptr;
    const float* gn_ada = nullptr;     // AdaGN table: scale = 1 + ada[c * stride + off], shift = ada[(Cout + c) * stride + off]
    int gn_ada_stride = 0;
    const int* gn_ada_idx = nullptr;   // [B] per-sample offset `off` into the table (null: 0)
    int gn_act = ACT_NONE, gn_groups = 32;
    float gn_eps = 1e-5f;
    void* gn_xch = nullptr;            // exchange words (conv_x3_gn_xch_bytes), one buffer per launch stream
    unsigned gn_tag = 0;               // != 0, unique per launch on this buffer
    int* gn_err = nullptr;             // raised (system scope) when a poll gives up
    // conv_x3 only, ragged batches: the launch's LIVE (sample, N tile) columns as a table (packed sample << 8 | N tile, samples counted
    // from cols_b0), so that the grid holds no workgroup that would exit at once: with per-sample column ranges in the id space and the
    // XCD-contiguous id order, the XCDs holding short samples ran out of tiles early (Model::register_cols builds it; null: all columns)
    const int* cols = nullptr;
    int ncols = 0, cols_b0 = 0;
    // conv_x3d only: act(y) of this conv written ALSO (or, y == null, only) as the split-precision planes the NEXT conv reads
    // ([B][next_c8][2][next_tp][8 fp16], next_halo zero columns on the left, live columns only: the consumer's zero margins are the caller's)
    void* next3 = nullptr;
    int next_c8 = 0, next_tp = 0, next_halo = 0, next_act = ACT_NONE;
    float next_slope = 0.f;
    int* next_sat = nullptr;       // raised when a scaled value leaves fp16's range (as launch_split_planes_ex)
    int ksplit = 1;
    int ksplit_max = 0;            // conv_x3: caller's cap on the split (0: the launcher's rule)
    int epi_vec = 0;               // conv_x3: y / res rows are 16-byte aligned -> LDS-staged epilogue with 16-byte stores (set by the launcher)
    float* kpart = nullptr;
    int* kcount = nullptr;
    int ablate = 0;                // experiments only (DTTS_CONV_ABLATE): 1 skip global loads, 2 skip LDS stores, 4 skip barriers
};

void launch_conv_gemm(const ConvParams& p, hipStream_t stream);

}  // namespace dtts
