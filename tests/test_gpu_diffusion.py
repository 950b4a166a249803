"""GPU parity: HIP diffusion stage (through the C ABI) vs the numpy oracle and the reference-made golden fixtures."""
import numpy as np
import pytest

from conftest import tol

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def rt(weights):
    from detail_tts_amd.runtime import Runtime
    return Runtime(weights, folded=True, parts=("diffusion",))


def test_philox_normal_matches_spec(rt):
    from oracle import philox
    z = host(rt.op_philox_normal(1001, 1234, [3, 9], philox.STAGE_DIFF_STEP, 17))
    for r, sid in enumerate((3, 9)):
        ref = philox.normal(1234, sid, philox.STAGE_DIFF_STEP, 17, 1001)
        assert maxabs(z[r], ref) < 2e-5


@pytest.mark.parametrize("cfg", [
    dict(cin=128, cout=768, k=3, pad=1, T=48),
    dict(cin=768, cout=768, k=1, pad=0, T=200),
    dict(cin=768, cout=256, k=3, pad=1, T=333),
    dict(cin=200, cout=200, k=11, pad=25, dil=5, T=300),
    dict(cin=100, cout=100, k=7, pad=9, dil=3, T=257),
    dict(cin=50, cout=50, k=3, pad=1, T=500),
    dict(cin=25, cout=25, k=11, pad=5, T=130),
    dict(cin=12, cout=1, k=7, pad=3, T=1000),
    dict(cin=128, cout=768, k=3, pad=1, stride=2, T=101),
])
def test_conv1d_kernel(cfg):
    from detail_tts_amd.packing import pack_conv
    from detail_tts_amd.runtime import Runtime
    from oracle import ops
    rs = np.random.RandomState(0)
    cin, cout, k, T = cfg["cin"], cfg["cout"], cfg["k"], cfg["T"]
    stride, dil, pad = cfg.get("stride", 1), cfg.get("dil", 1), cfg["pad"]
    w = (rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    wp, bp = pack_conv(w, b)
    r = Runtime({}, parts=(), extra={"t.wp": wp, "t.bp": bp})
    x = rs.randn(2, cin, T).astype(np.float32)
    lens = [T, max(1, T - 37)]
    y = host(r.op_conv1d("t", dev(x), cout, k, stride=stride, dil=dil, pad=pad, lens_in=lens))
    for bi, L in enumerate(lens):
        ref = ops.conv1d(x[bi:bi + 1, :, :L], w, b, stride=stride, padding=pad, dilation=dil)[0]
        assert maxabs(y[bi, :, :ref.shape[1]], ref) < 2e-5, (cfg, bi)


def test_conv_transpose_as_phases():
    from detail_tts_amd.packing import convtranspose_as_phases, pack_conv
    from detail_tts_amd.runtime import Runtime
    from oracle import ops
    rs = np.random.RandomState(1)
    for (cin, cout, k, s, p, T) in [(400, 200, 16, 8, 4, 48), (200, 100, 8, 4, 2, 100), (100, 50, 2, 2, 0, 333), (25, 12, 2, 2, 0, 64)]:
        w = (rs.randn(cin, cout, k) / np.sqrt(cin)).astype(np.float32)
        b = rs.randn(cout).astype(np.float32)
        weq, pad = convtranspose_as_phases(w, s, p)
        wp, bp = pack_conv(weq, np.tile(b, s))
        r = Runtime({}, parts=(), extra={"t.wp": wp, "t.bp": bp})
        x = rs.randn(2, cin, T).astype(np.float32)
        y = host(r.op_conv1d("t", dev(x), cout, weq.shape[2], pad=pad, phases=s))
        ref = ops.conv_transpose1d(x, w, b, stride=s, padding=p)
        assert y.shape == ref.shape
        assert maxabs(y, ref) < 2e-5, (cin, cout, k, s)


def test_gated_conv_epilogue():
    from detail_tts_amd.packing import gate_perm, pack_conv
    from detail_tts_amd.runtime import Runtime
    from oracle import ops
    rs = np.random.RandomState(2)
    hid, T = 192, 150
    w = (rs.randn(2 * hid, hid, 5) / np.sqrt(hid * 5)).astype(np.float32)
    b = rs.randn(2 * hid).astype(np.float32) * 0.1
    wp, bp = pack_conv(w, b, row_perm=gate_perm(2 * hid))
    r = Runtime({}, parts=(), extra={"t.wp": wp, "t.bp": bp})
    x = rs.randn(2, hid, T).astype(np.float32)
    a = ops.conv1d(x, w, b, padding=2)
    for gate, ref in ((1, np.tanh(a[:, :hid]) * ops.sigmoid(a[:, hid:])), (2, a[:, :hid] * ops.sigmoid(a[:, hid:]))):
        y = host(r.op_conv1d("t", dev(x), 2 * hid, 5, pad=2, gate=gate))
        assert maxabs(y, ref) < 2e-5


@pytest.mark.parametrize("T,lens", [(48, None), (200, [200, 131]), (333, [64, 333])])
def test_attention_block(rt, weights, T, lens):
    from oracle import diffusion as D
    rs = np.random.RandomState(3)
    B = 2
    x = rs.randn(B, 768, T).astype(np.float32)
    p = "diffusion.layers.3.attn"
    y = host(rt.op_attention_block(p, dev(x), lens))
    for b in range(B):
        L = T if lens is None else lens[b]
        ref = D.attention_block(weights, p, x[b:b + 1, :, :L], 16)[0]
        assert maxabs(y[b, :, :L], ref) < 1e-4, (T, b)


def test_attention_block_with_growing_scores_rescales_mid_sequence(rt, weights):
    """The trunk attention's lazy running maximum: with keys whose scores grow along the sequence a query raises its maximum (and the
    kernel rescales its accumulators - an asm block behind MFMA results, csrc/attention_x3b.hip) at many blocks, not only at the first
    one where the accumulators are still zero (the only rescale the small random fixtures ever take).  A ragged batch (band, far and
    masked-tail blocks in one launch) against the dense oracle.  (Round 5's first, block-skewed build read MFMA results too early in
    that asm and failed only the T = 936 forward; in the shipped three-block pipeline the rescale runs a whole step after the MFMAs it
    follows - a build with its wait states removed passes this test and the forward - so the padding there is belt and braces.)"""
    from oracle import diffusion as D
    rs = np.random.RandomState(5)
    T, lens = 700, [700, 413, 90]
    ramp = (0.3 + 2.7 * np.arange(T) / T).astype(np.float32)            # later frames are ~9 x larger: q.k grows ~80 x along the keys
    x = (rs.randn(3, 768, T) * ramp[None, None, :]).astype(np.float32)
    p = "diffusion.layers.5.attn"
    y = host(rt.op_attention_block(p, dev(x), lens))
    for b, L in enumerate(lens):
        ref = D.attention_block(weights, p, x[b:b + 1, :, :L], 16)[0]
        assert maxabs(y[b, :, :L], ref) < 2e-4 * max(1.0, float(np.abs(ref).max())), (b, maxabs(y[b, :, :L], ref), float(np.abs(ref).max()))


@pytest.mark.parametrize("S", [2, 3, 4])
def test_attention_key_split_of_small_launches_vs_unsplit_and_oracle(rt, weights, S):
    """Round 6: launches of <= 2 samples cut the keys of every (sample, head, 128-query block) into S ranges of
    whole 64-key tiles, one workgroup each; the last wave to arrive merges the (O, l, m) partials in split order.  Against the unsplit
    kernel (fp32 summation-order noise only) and against the dense oracle: a ragged pair whose short row has FEWER tiles than splits
    (empty ranges), lengths on / off tile edges, growing scores (the lazy maximum differs between ranges), repeated (the arrival
    counters are back at zero after every launch; the merge never depends on who arrives last: bit-identical)."""
    from oracle import diffusion as D
    rs = np.random.RandomState(50 + S)
    T, lens = 700, [700, 40]
    ramp = (0.3 + 2.7 * np.arange(T) / T).astype(np.float32)
    x = (rs.randn(2, 768, T) * ramp[None, None, :]).astype(np.float32)
    p = "diffusion.layers.5.attn"
    try:
        rt.set_option("attn_ksplit_cus", 1 << 20)          # (by default a launch is split only while its workgroups x S find a CU each)
        rt.set_option("attn_ksplit", 1)
        y1 = host(rt.op_attention_block(p, dev(x), lens))
        rt.set_option("attn_ksplit", S)
        ys = host(rt.op_attention_block(p, dev(x), lens))
        again = [host(rt.op_attention_block(p, dev(x), lens)) for _ in range(3)]
        one = host(rt.op_attention_block(p, dev(x[:1, :, :576]), [512]))
        rt.set_option("attn_ksplit", 1)
        one1 = host(rt.op_attention_block(p, dev(x[:1, :, :576]), [512]))
    finally:
        rt.set_option("attn_ksplit", 4)
        rt.set_option("attn_ksplit_cus", 256)
    assert all(np.array_equal(a, ys) for a in again)
    for b, L in enumerate(lens):
        ref = D.attention_block(weights, p, x[b:b + 1, :, :L], 16)[0]
        scale = max(1.0, float(np.abs(ref).max()))
        assert maxabs(ys[b, :, :L], ref) < 2e-4 * scale, (b, maxabs(ys[b, :, :L], ref))
        tol(f"attn_ksplit{S}_vs_unsplit_row{b}", maxabs(ys[b, :, :L], y1[b, :, :L]) / scale, 2e-5)
    assert float(np.abs(ys - y1).max()) > 0.0          # the split path did run
    tol(f"attn_ksplit{S}_single_sample_vs_unsplit", maxabs(one[0, :, :512], one1[0, :, :512]) / max(1.0, float(np.abs(one1).max())), 2e-5)


def test_attention_block_1536(rt, weights):
    from oracle import diffusion as D
    rs = np.random.RandomState(4)
    x = rs.randn(1, 1536, 70).astype(np.float32)
    p = "diffusion.contextual_embedder.4"
    y = host(rt.op_attention_block(p, dev(x)))
    assert maxabs(y, D.attention_block(weights, p, x, 16)) < 1e-4


def test_resblock(rt, weights):
    from oracle import diffusion as D
    rs = np.random.RandomState(5)
    sched = D.make_schedule()
    x = rs.randn(2, 768, 100).astype(np.float32)
    lens = [100, 77]
    for prefix, step in (("diffusion.layers.0.resblk", 49), ("diffusion.layers.11", 3), ("diffusion.conditioning_timestep_integrator.1.resblk", 20)):
        y = host(rt.op_resblock(prefix, dev(x), step, lens))
        temb = D.time_embed(weights, [sched["timestep_map"][step]], 768)
        for b, L in enumerate(lens):
            ref = D.res_block(weights, prefix, x[b:b + 1, :, :L], temb)[0]
            assert maxabs(y[b, :, :L], ref) < 1e-4, (prefix, b)


def test_conditioning_and_code_emb_golden(rt, golden):
    g = golden("diff_cond")
    cond = host(rt.diff_conditioning(dev(g["refer"])))
    assert maxabs(cond, g["cond_latent"]) < 1e-4
    lat_cm = np.ascontiguousarray(g["latent"].transpose(0, 2, 1))
    ce = host(rt.diff_timestep_independent(dev(lat_cm), dev(g["cond_latent"])))
    assert maxabs(ce, g["code_emb"]) < 2e-4


def test_conditioning_varlen_batch(rt, weights):
    from oracle import diffusion as D
    rs = np.random.RandomState(6)
    refer = (rs.randn(2, 128, 90) * 2 - 5).astype(np.float32)
    lens = [90, 61]
    cond = host(rt.diff_conditioning(dev(refer), lens))
    for b, L in enumerate(lens):
        assert maxabs(cond[b], D.get_conditioning(weights, refer[b:b + 1, :, :L])[0]) < 1e-4
    lat = rs.randn(2, 768, 20).astype(np.float32)
    ln = [20, 13]
    ce = host(rt.diff_timestep_independent(dev(lat), dev(cond), ln))
    for b, L in enumerate(ln):
        ref = D.timestep_independent(weights, lat[b:b + 1, :, :L].transpose(0, 2, 1), cond[b:b + 1], 4 * L)[0]
        assert maxabs(ce[b, :, :4 * L], ref) < 2e-4


def test_diffusion_forward_golden(rt, golden):
    g = golden("diff_forward")
    sched_step = 47                                  # timestep_map[47] == 3836
    assert int(g["ts"][0]) == 3836
    oc = host(rt.diff_forward(dev(g["x"]), sched_step, dev(g["code_emb"])))
    ou = host(rt.diff_forward(dev(g["x"]), sched_step, cond_free=True))
    assert maxabs(oc, g["out_cond"]) < 3e-4, maxabs(oc, g["out_cond"])
    assert maxabs(ou, g["out_uncond"]) < 3e-4


def test_sampler_steps_golden(rt, golden):
    """3 ancestral steps from the reference's x_T with Philox noise generated ON DEVICE."""
    g = golden("diff_sampler_steps")
    x = rt.diff_sample(dev(g["code_emb"]), int(g["seed"]), [int(g["sample_id"])], n_steps=3, denorm=False)
    ref = g["x_after_47"]
    err = maxabs(host(x), ref)
    # eps errors are amplified 153x before the clamp at i=49: compare in RMS too
    rms = float(np.sqrt(np.mean((host(x) - ref) ** 2)))
    assert rms < 2e-3 and err < 5e-2, (rms, err)
    x1 = rt.diff_sample(dev(g["code_emb"]), int(g["seed"]), [int(g["sample_id"])], n_steps=1, denorm=False)
    assert maxabs(host(x1), g["x_after_49"]) < 2e-2


def test_sampler_varlen_batch_equals_single(rt, weights):
    """A padded 2-utterance batch must reproduce each utterance run alone (SURVEY §0 'batch 8' row)."""
    rs = np.random.RandomState(7)
    ce = rs.randn(2, 768, 64).astype(np.float32)
    lens = [64, 40]
    xb = host(rt.diff_sample(dev(ce), 99, [11, 12], lens=lens, n_steps=2, denorm=True))
    for b, L in enumerate(lens):
        xs = host(rt.diff_sample(dev(ce[b:b + 1, :, :L]), 99, [11 + b], n_steps=2, denorm=True))
        assert maxabs(xb[b, :, :L], xs[0]) < 1e-4, b


def test_tiny_and_ragged_lengths(rt, weights):
    """T = 4 (one code), and lengths that are not multiples of any tile."""
    from oracle import diffusion as D
    rs = np.random.RandomState(8)
    sched = D.make_schedule()
    for T in (4, 68, 132):
        x = rs.randn(1, 128, T).astype(np.float32)
        ce = rs.randn(1, 768, T).astype(np.float32)
        out = host(rt.diff_forward(dev(x), 33, dev(ce)))
        ref = D.diffusion_forward(weights, x, [sched["timestep_map"][33]], ce)
        assert maxabs(out, ref) < 3e-4, T


@pytest.mark.parametrize("lens", [[600, 192, 385, 40, 577, 384], [600, 100, 384], [192, 600]])
def test_ragged_batch_launching_only_live_columns_is_bit_identical(rt, weights, lens):
    """Option conv_cols (default on): the trunk convs of a ragged batch launch one workgroup per live (sample, N tile) column from a
    table built of the host lengths, instead of a grid over the padded length.  Same tiles, same arithmetic: the sampler's output
    (integrator precompute + 2 CFG steps, both chunkings of the cond | uncond stack) must not change by one bit - and short
    samples that end exactly on a tile boundary (192, 384), inside the first tile (40) and one column past a boundary (385, 577)
    must keep every live column."""
    rs = np.random.RandomState(31)
    B, T = len(lens), max(lens)
    ce = dev(rs.randn(B, 768, T) * 0.5)
    outs = {}
    try:
        for flag in (0, 1):
            rt.set_option("conv_cols", flag)
            outs[flag] = host(rt.diff_sample(ce, 5, list(range(B)), lens=lens, n_steps=2, denorm=True))
    finally:
        rt.set_option("conv_cols", 1)
    for b, L in enumerate(lens):
        assert np.isfinite(outs[1][b, :, :L]).all()
        assert float(np.abs(outs[1][b, :, :L]).max()) > 0.1
        assert np.array_equal(outs[0][b, :, :L], outs[1][b, :, :L]), b
    # and against each sample alone (no table: a single sample has no dead column)
    b = int(np.argmin(lens))
    alone = host(rt.diff_sample(ce[b:b + 1, :, :lens[b]].contiguous(), 5, [b], n_steps=2, denorm=True))
    assert maxabs(outs[1][b, :, :lens[b]], alone[0]) < 1e-4


@pytest.mark.parametrize("lens", [[333], [132, 77, 200]])
def test_integrator_chunks_under_the_sampling_loop_are_bit_identical(rt, weights, lens):
    """Option integ_pipeline (off by default, DESIGN.md par. 4.5): only the first chunk of the conditioning_timestep_integrator's step outputs
    (vqvae/diff_model.py:295; it never sees x_t) is evaluated in front of the sampling loop, the later chunks run on a low-priority
    stream under the first steps, which wait for a chunk's event at its first step.  Same launches on the same inputs: all 50 steps
    (3 chunks at batch 1, 9 at a ragged batch of 3) must not change by one bit, also when the call is repeated (the chunks' scratch
    is reused by the next call)."""
    rs = np.random.RandomState(41)
    B, T = len(lens), max(lens)
    ce = dev(rs.randn(B, 768, T) * 0.5)
    outs = {}
    try:
        for flag in (0, 1, 1):
            rt.set_option("integ_pipeline", flag)
            outs.setdefault(flag, []).append(host(rt.diff_sample(ce, 6, list(range(B)), lens=lens, n_steps=50, denorm=True)))
    finally:
        rt.set_option("integ_pipeline", 0)
    rt.set_option("integ_pipeline", -1)                 # by batch size: on for these calls
    try:
        dflt = host(rt.diff_sample(ce, 6, list(range(B)), lens=lens, n_steps=50, denorm=True))
    finally:
        rt.set_option("integ_pipeline", 0)
    dflt0 = host(rt.diff_sample(ce, 6, list(range(B)), lens=lens, n_steps=50, denorm=True))
    for b, L in enumerate(lens):
        ref = outs[0][0][b, :, :L]
        assert np.isfinite(ref).all() and float(np.abs(ref).max()) > 0.1
        for o in outs[1] + [dflt, dflt0]:
            assert np.array_equal(ref, o[b, :, :L]), b


def test_errors_are_reported_not_crashes(rt):
    from detail_tts_amd.runtime import DttsError
    x = torch.zeros(1, 128, 8, device="cuda")
    with pytest.raises(DttsError):
        rt.diff_forward(x, 50, torch.zeros(1, 768, 8, device="cuda"))          # step out of range
    with pytest.raises(DttsError):
        rt.diff_forward(x.cpu(), 1, None)                                        # host tensor
    with pytest.raises(DttsError):
        rt.op_attention_block("diffusion.layers.99.attn", torch.zeros(1, 768, 8, device="cuda"))


def test_split_precision_path_equals_fp32_path(rt):
    """The trunk's default kernels compute every fp32 product as three fp16 MFMA products (two fp16 planes per operand) (3 x bf16 split operands, fp32
    accumulate).  They must agree with the exact fp32-MFMA kernels (option conv_x3 = 0) to fp32 rounding, on a whole
    DiffusionTts.forward at a ragged length (T = 333 is not a multiple of any tile size)."""
    rs = np.random.RandomState(21)
    B, T = 2, 333
    x = dev(rs.randn(B, 128, T))
    code_emb = dev(rs.randn(B, 768, T) * 0.5)
    lens = [333, 170]
    outs = {}
    for flag in (1, 0):
        rt.set_option("conv_x3", flag)
        outs[flag] = host(rt.diff_forward(x, 17, code_emb, lens=lens))
    rt.set_option("conv_x3", 1)
    for b, L in enumerate(lens):
        a, r = outs[1][b, :, :L], outs[0][b, :, :L]
        assert float(np.abs(r).max()) > 0.1
        assert maxabs(a, r) < 5e-5 * max(1.0, float(np.abs(r).max())), (b, maxabs(a, r))
        rel = float(np.sqrt(np.mean((a - r) ** 2)) / np.sqrt(np.mean(r ** 2)))
        assert rel < 1e-5, rel          # measured 2.7e-6 through ~100 layers: the spread of two fp32 summation orders


def test_trunk_blocks_random_shapes(rt, weights):
    """Split-precision ResBlock / AttentionBlock at random ragged shapes (tile edges: T around 64 / 128 multiples, very short rows)."""
    from oracle import diffusion as D
    rs = np.random.RandomState(99)
    sched = D.make_schedule()
    for T, lens in ((5, [5, 3]), (63, [63, 17]), (65, [65, 64]), (127, [100, 127]), (129, [129, 1]), (257, [256, 130])):
        B = len(lens)
        x = rs.randn(B, 768, T).astype(np.float32)
        step = int(rs.randint(0, 50))
        y = host(rt.op_resblock("diffusion.layers.5.resblk", dev(x), step, lens))
        z = host(rt.op_attention_block("diffusion.layers.5.attn", dev(x), lens))
        temb = D.time_embed(weights, [sched["timestep_map"][step]], 768)
        for b, L in enumerate(lens):
            ref = D.res_block(weights, "diffusion.layers.5.resblk", x[b:b + 1, :, :L], temb)[0]
            assert maxabs(y[b, :, :L], ref) < 1e-4, ("resblock", T, b)
            ref = D.attention_block(weights, "diffusion.layers.5.attn", x[b:b + 1, :, :L], 16)[0]
            assert maxabs(z[b, :, :L], ref) < 1e-4, ("attention", T, b)


def test_small_launch_split_k_is_deterministic_under_load(rt):
    """Small launches split their K range over several workgroups that meet through an arrival counter; the sums are taken in split
    order, so repeated runs are bit-identical - also while another stream keeps the chip busy."""
    rs = np.random.RandomState(11)
    x = dev(rs.randn(2, 768, 100).astype(np.float32))
    big = torch.randn(4096, 4096, device="cuda")
    ref = rt.op_resblock("diffusion.layers.2.resblk", x, 9, [100, 71])
    ref_a = rt.op_attention_block("diffusion.layers.2.attn", ref, [100, 71])
    side = torch.cuda.Stream()
    for i in range(40):
        if i % 4 == 0:                      # unrelated work on another stream (a handle's own entry points are one-at-a-time per stage)
            with torch.cuda.stream(side):
                big = torch.tanh(big @ big) * 0.01
        y = rt.op_resblock("diffusion.layers.2.resblk", x, 9, [100, 71])
        a = rt.op_attention_block("diffusion.layers.2.attn", y, [100, 71])
        assert torch.equal(y, ref) and torch.equal(a, ref_a), i
    torch.cuda.synchronize()
