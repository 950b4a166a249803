// fp32 MFMA conv-as-GEMM for gfx950 (see conv_gemm.h).
//
// Tiling: a 256-thread workgroup (4 waves) owns a BM x BN tile of y[b]; each wave owns a
// (BM/WGM) x (BN/WGN) sub-tile built from 32x32 v_mfma_f32_32x32x2_f32 tiles (exact fp32,
// 64 FLOP/clk/SIMD — the only fp32 matrix path on gfx950).  K runs over (ci-block of 16, tap):
// the input tile [16][XW] (XW = (BN-1)*stride + (KW-1)*dil + 1, halo included) is staged in
// LDS ONCE per ci-block and re-used by every tap through a shifted read; the weight tile
// [16][BM] is staged per (ci-block, tap).  Both tiles are K-major in LDS so every MFMA operand
// read is a conflict-free ds_read_b32 of 32 consecutive floats per lane-half.  Global->LDS goes
// through registers (double-buffered LDS, one barrier per K-step) because the input path applies
// the fused prologue (GroupNorm affine + SiLU / leaky-relu, zero padding, per-sample length).
#include <cstdlib>
#include "conv_gemm_kernel.h"

namespace dtts {

// Tile choice.  DTTS_CONV_BN / DTTS_CONV_BK (environment) force a shape for experiments.
static int env_int(const char* name) {
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
}

void launch_conv_gemm(const ConvParams& p_in, hipStream_t stream) {
    static const int ablate = env_int("DTTS_CONV_ABLATE");
    ConvParams p = p_in;
    p.ablate = ablate;
    DTTS_REQUIRE(p.B > 0 && p.Nout > 0 && p.Cout > 0 && p.Cin > 0, "empty conv");
    DTTS_REQUIRE(p.x && p.w && p.y, "null pointer");
    DTTS_REQUIRE(p.gate == GATE_NONE || p.phases == 1, "gate+phases unsupported");
    static const int force_bn = env_int("DTTS_CONV_BN"), force_bk = env_int("DTTS_CONV_BK");
    const int halo = (p.KW - 1) * p.dil;
    auto fits = [&](int bn) { return (bn - 1) * p.stride + halo + 1 <= XW_MAX; };
    const bool bk32 = (p.CinP % 32 == 0) && force_bk == 32 && p.KW * p.dil <= 3;   // measured slower than BK=16: opt-in only
    // Small launches (GPT prefill, conditioning encoders, the WaveNets' 1x1 convs: a few dozen 128 x 128 tiles on 256 CUs): a tile's K loop
    // is a serial chain of 64-cycle fp32 MFMAs, so the launch takes one tile's time however empty the chip is - 64 x 64 tiles cut that
    // chain to a quarter per K-step on four times the workgroups (same k order per output: identical sums).  DTTS_CONV_SMALL_TILES = n:
    // launches of at most n 128 x 128 tiles take the small tile (0: never).
    static const int small_tiles = []() { const char* v = getenv("DTTS_CONV_SMALL_TILES"); return v ? atoi(v) : 384; }();
    if (small_tiles > 0 && p.CoutP % 64 == 0 && fits(64) && (long long)cdiv(p.CoutP, 128) * cdiv(p.Nout, 128) * p.B <= small_tiles) {
        // ... and a small launch is bound by one memory latency per K-step (two-stage pipeline): 32-channel steps halve their number
        static const int small_bk = []() { const char* v = getenv("DTTS_CONV_SMALL_BK"); return v ? atoi(v) : 32; }();
        // (64-channel steps: 66 KiB of LDS, measured slower again - prefill 6.4 vs 5.5 ms)
        if (small_bk == 32 && p.CinP % 32 == 0 && p.KW * p.dil <= 3) launch_conv_tile<64, 64, 2, 2, 32>(p, stream, "conv_gemm_kernel<64,64,k32>");
        else launch_conv_tile<64, 64, 2, 2, 16>(p, stream, "conv_gemm_kernel<64,64,k16>");
        return;
    }
    if (p.CoutP % 128 == 0) {
        // BN=128 has the better MFMA:staging ratio; take it unless the ragged tail wastes more than ~10 % of the columns
        const int pad128 = round_up(p.Nout, 128), pad64 = round_up(p.Nout, 64);
        bool wide = fits(128) && (pad128 - pad64) * 10 <= p.Nout;
        if (force_bn == 128 && fits(128)) wide = true;
        if (force_bn == 64) wide = false;
        if (wide) {
            if (bk32) launch_conv_tile<128, 128, 2, 2, 32>(p, stream, "conv_gemm_kernel<128,128,k32>");
            else launch_conv_tile<128, 128, 2, 2, 16>(p, stream, "conv_gemm_kernel<128,128,k16>");
        } else {
            DTTS_REQUIRE(fits(64), "conv halo too large");
            if (bk32) launch_conv_tile<128, 64, 2, 2, 32>(p, stream, "conv_gemm_kernel<128,64,k32>");
            else launch_conv_tile<128, 64, 2, 2, 16>(p, stream, "conv_gemm_kernel<128,64,k16>");
        }
    } else if (p.CoutP % 64 == 0) {
        if (fits(128) && p.Nout > 64) launch_conv_tile<64, 128, 2, 2, 16>(p, stream, "conv_gemm_kernel<64,128,k16>");
        else { DTTS_REQUIRE(fits(64), "conv halo too large"); launch_conv_tile<64, 64, 2, 2, 16>(p, stream, "conv_gemm_kernel<64,64,k16>"); }
    } else {
        DTTS_REQUIRE(p.CoutP % 32 == 0, "CoutP must be a multiple of 32");
        DTTS_REQUIRE(fits(128), "conv halo too large");
        launch_conv_tile<32, 128, 1, 4, 16>(p, stream, "conv_gemm_kernel<32,128,k16>");
    }
}

}  // namespace dtts
