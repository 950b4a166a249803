"""GPU parity at the HEADLINE sizes (BASELINE.json configs[1]/[2]/[4]): T = 936 mel frames, 234 codes, batch 8, and the 60 s
attention length T = 5624.  Two checkers: `tests/golden/fullsize.npz` = subsampled outputs of the REFERENCE ITSELF at these sizes
(make_golden_fullsize.py), and the numpy oracle densely (every element) on the same inputs.  Everything goes through the C ABI."""
import numpy as np
import pytest

from conftest import tol
from fullsize_inputs import N_CODES, T, inputs, sub

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def I():
    return inputs()


@pytest.fixture(scope="module")
def G(golden):
    return golden("fullsize")


@pytest.fixture(scope="module")
def rt(weights):
    from detail_tts_amd.runtime import Runtime
    return Runtime(weights, folded=True, parts=("diffusion", "gpt", "vocoder"))


def check_sub(out, G, key, tol, ch_stride=None):
    s, t = sub(out, G, ch_stride)
    e = max(maxabs(s, G[key + "_s"]), maxabs(t, G[key + "_t"]))
    assert e < tol, (key, e)
    return e


@pytest.mark.parametrize("step", [47, 0])
@pytest.mark.parametrize("x3", [1, 0])
def test_diffusion_forward_T936_vs_reference(rt, I, G, step, x3):
    """DiffusionTts.forward (vqvae/diff_model.py:262-322) at T = 936 (8 N-tiles, 15 key tiles): split-precision AND exact-fp32
    kernels vs the reference's own output, cond + uncond, first and last sampling step."""
    rt.set_option("conv_x3", x3)
    try:
        oc = host(rt.diff_forward(dev(I["x"]), step, dev(I["code_emb"])))[0]
        ou = host(rt.diff_forward(dev(I["x"]), step, cond_free=True))[0]
    finally:
        rt.set_option("conv_x3", 1)
    check_sub(oc, G, f"fwd{step}_cond", 3e-4)
    check_sub(ou, G, f"fwd{step}_uncond", 3e-4)


def test_diffusion_forward_T936_batch2_vs_oracle_dense(rt, weights, I):
    """B = 2 (one full-length row, one ragged row) at step 47: EVERY output element vs the oracle, cond and uncond."""
    from oracle import diffusion as D
    sched = D.make_schedule()
    rs = np.random.RandomState(12)
    lens = [T, 871]
    x = np.concatenate([I["x"], rs.randn(1, 128, T).astype(np.float32)])
    ce = np.concatenate([I["code_emb"], (rs.randn(1, 768, T) * 0.5).astype(np.float32)])
    oc = host(rt.diff_forward(dev(x), 47, dev(ce), lens=lens))
    ou = host(rt.diff_forward(dev(x), 47, cond_free=True, lens=lens))
    ts = [sched["timestep_map"][47]]
    for b, L in enumerate(lens):
        rc = D.diffusion_forward(weights, x[b:b + 1, :, :L], ts, ce[b:b + 1, :, :L])[0]
        ru = D.diffusion_forward(weights, x[b:b + 1, :, :L], ts, conditioning_free=True)[0]
        assert maxabs(oc[b, :, :L], rc) < 3e-4, (b, maxabs(oc[b, :, :L], rc))
        assert maxabs(ou[b, :, :L], ru) < 3e-4, (b, maxabs(ou[b, :, :L], ru))


def test_fused_groupnorm_forward_B6_T936_vs_reference_and_unfused(rt, I, G):
    """The trunk with every GroupNorm + activation + split folded into the epilogue of the conv in front of it (conv_x3.h "fused
    GroupNorm": tiles exchange partial statistics through tagged words; batches of >= 5 utterances at T <= 1152) against (a) the
    reference's own DiffusionTts.forward for row 0 and (b) the launch-by-launch GroupNorm passes for every row of a RAGGED batch of 6
    (lengths that end inside a tile, on a tile edge - the zero halo column of the next tile - and one full row).  Only the order of
    the statistics' fp32 sums differs between (a) and (b)."""
    rs = np.random.RandomState(13)
    B = 6
    lens = [T, 800, T, 577, 768, 192]
    x = np.concatenate([I["x"], rs.randn(B - 1, 128, T).astype(np.float32)])
    ce = np.concatenate([I["code_emb"], (rs.randn(B - 1, 768, T) * 0.5).astype(np.float32)])
    outs = {}
    for fuse in (1, 0):
        rt.set_option("gn_fuse", fuse)
        try:
            outs[fuse] = (host(rt.diff_forward(dev(x), 47, dev(ce), lens=lens)), host(rt.diff_forward(dev(x), 47, cond_free=True, lens=lens)))
        finally:
            rt.set_option("gn_fuse", 0)
    check_sub(outs[1][0][0], G, "fwd47_cond", 3e-4)
    check_sub(outs[1][1][0], G, "fwd47_uncond", 3e-4)
    for h in (0, 1):
        for b, L in enumerate(lens):
            tol(f"fused_gn_vs_unfused_half{h}_row{b}", maxabs(outs[1][h][b, :, :L], outs[0][h][b, :, :L]), 5e-5)
    # the fused path really ran: a repeated call is bit-identical (deterministic combination order)
    rt.set_option("gn_fuse", 1)
    try:
        again = host(rt.diff_forward(dev(x), 47, dev(ce), lens=lens))
    finally:
        rt.set_option("gn_fuse", 0)
    assert np.array_equal(again, outs[1][0]) and not np.array_equal(outs[1][0], outs[0][0])


@pytest.mark.parametrize("B", [1, 2])
def test_fused_groupnorm_on_split_k_launches_vs_reference_and_unfused(rt, I, G, B):
    """Batches 1 and 2 (launches of <= 128 tiles: split-K, 2 - 4 workgroups per tile): since round 5 the fused GroupNorm epilogue runs in
    the tile's reducing workgroup (option gn_fuse; off by default: measured slower there too).  Row 0 against the reference's own
    DiffusionTts.forward (vqvae/diff_model.py:262-322), every row (one ragged) against the separate passes."""
    rs = np.random.RandomState(14)
    lens = [T, 700][:B]
    x = np.concatenate([I["x"], rs.randn(1, 128, T).astype(np.float32)])[:B]
    ce = np.concatenate([I["code_emb"], (rs.randn(1, 768, T) * 0.5).astype(np.float32)])[:B]
    outs = {}
    for fuse in (1, 0):
        rt.set_option("gn_fuse", fuse)
        try:
            outs[fuse] = (host(rt.diff_forward(dev(x), 47, dev(ce), lens=lens)), host(rt.diff_forward(dev(x), 47, cond_free=True, lens=lens)))
        finally:
            rt.set_option("gn_fuse", 0)
    check_sub(outs[1][0][0], G, "fwd47_cond", 3e-4)
    check_sub(outs[1][1][0], G, "fwd47_uncond", 3e-4)
    for h in (0, 1):
        for b, L in enumerate(lens):
            tol(f"fused_gn_splitk_B{B}_vs_unfused_half{h}_row{b}", maxabs(outs[1][h][b, :, :L], outs[0][h][b, :, :L]), 5e-5)
    assert not np.array_equal(outs[1][0], outs[0][0])


def test_p_sample_T936_vs_reference(rt, I, G):
    """One GaussianDiffusion.p_sample (vqvae/utils/diffusion.py:445-485) at i = 49, T = 936, Philox noise on the device."""
    x1, x0 = rt.diff_p_sample(dev(I["x"]), dev(I["code_emb"]), 49, 1234, [2], return_x0=True)
    # eps errors are amplified 153x into pred_xstart at i = 49 before the clamp
    check_sub(host(x0)[0], G, "ps49_x0", 5e-3)
    check_sub(host(x1)[0], G, "ps49_x", 1e-3)


def test_attention_block_T936_and_T5624_vs_reference(rt, weights, I, G):
    """AttentionBlock (vqvae/utils/diff_util.py:146-169, 209-215) at the 10 s and the 60 s (configs[4]) lengths."""
    from oracle import diffusion as D
    p = "diffusion.layers.3.attn"
    y = host(rt.op_attention_block(p, dev(I["xa"])))[0]
    check_sub(y, G, "attn", 1e-4)
    assert maxabs(y, D.attention_block(weights, p, I["xa"], 16)[0]) < 1e-4          # dense, every element
    yl = host(rt.op_attention_block(p, dev(I["xa_long"])))[0]
    check_sub(yl, G, "attn_long", 1e-4, ch_stride=16)
    # ragged batch at the long length: rows of a batch equal the rows alone
    rs = np.random.RandomState(13)
    xb = np.concatenate([I["xa_long"], rs.randn(1, 768, 5624).astype(np.float32)])
    yb = host(rt.op_attention_block(p, dev(xb), [5624, 5001]))
    assert maxabs(yb[0], yl) < 1e-6
    alone = host(rt.op_attention_block(p, dev(xb[1:2, :, :5001])))[0]
    assert maxabs(yb[1, :, :5001], alone) < 1e-6


def test_vocoder_T936_vs_reference_and_oracle(rt, weights, I, G):
    """infer_flowvae (vqvae/model_24k.py:848-863) at T = 936: z and the 24 kHz waveform vs the reference; waveform vs the oracle."""
    from oracle import vocoder as V
    wav, z = rt.vocoder(dev(I["mel"]), 1234, [3], return_z=True)
    wav, z = host(wav)[0, 0], host(z)[0]
    check_sub(z, G, "voc_z", 2e-4)
    ref_rms = float(G["voc_wav_rms"])
    # seed-0 weights: the waveform is 7e-3 RMS of which only 1e-3 is driven by z, so the limit sits at ~20 x the measured error
    # (1e-8 class), not at north_star's 1e-3; the "signal" weight set below makes the same comparison on a waveform that is 6/7 z-driven
    tol("voc936_wav_sub_rms", rms(wav[::97], G["voc_wav_s"]), 1e-7)
    tol("voc936_wav_tail_rms", rms(wav[-2048:], G["voc_wav_t"]), 1e-7)
    assert ref_rms > 1e4 * rms(wav[::97], G["voc_wav_s"])
    ref = V.infer_flowvae(weights, I["mel"], [T], 1234, [3])
    tol("voc936_wav_vs_oracle_rms", rms(wav, np.asarray(ref).reshape(-1)), 1e-7)


def test_gpt_decode_latents_234_codes_vs_reference(rt, I, G):
    """KV-cache decode over 234 forced codes (L grows to ~300 keys): the per-step latents equal the reference's
    UnifiedVoice.forward(return_latent=True) values, at the LAST positions too; teacher-forced pass alike."""
    codes, ncodes, lat = rt.gpt_generate(dev(I["refer"]), None, [I["text"][0]], 1, [0], max_generate_length=N_CODES + 1,
                                         forced_codes=[I["codes"][0]])
    assert np.array_equal(codes[0, :N_CODES], I["codes"][0]) and ncodes[0] == N_CODES + 1
    lat = host(lat)[0].T                                   # [G, 768]
    assert maxabs(lat[:N_CODES][::9], G["gpt_lat_s"]) < 2e-4, maxabs(lat[:N_CODES][::9], G["gpt_lat_s"])
    assert maxabs(lat[N_CODES - 8:N_CODES], G["gpt_lat_t"]) < 2e-4
    tf = host(rt.gpt_latents(dev(I["refer"]), None, [I["text"][0]], [I["codes"][0]]))[0].T
    assert maxabs(tf[::9], G["gpt_lat_s"]) < 2e-4 and maxabs(tf[-8:], G["gpt_lat_t"]) < 2e-4


def test_gpt_graph_replay_equals_eager_and_session_api(rt, I):
    """dtts_gpt_prefill / _decode (captured hipGraphs) / _decode_step (eager launches) / _finish: same codes and latents whichever
    way the steps are issued; steps past max_generate_length are no-ops."""
    B, Gn = 3, 41
    rs = np.random.RandomState(14)
    rt.set_option("gpt_graph", 1)
    refer = dev((rs.randn(B, 128, 200) * 2 - 5).astype(np.float32))
    texts = [np.concatenate([rs.randint(3, 255, n), [0]]) for n in (12, 30, 7)]
    kw = dict(max_generate_length=Gn, suppress_eos=True)
    ref_codes, ref_n, ref_lat = rt.gpt_generate(refer, [200, 150, 90], texts, 5, [1, 2, 3], **kw)
    for mode in ("eager", "graph", "mixed"):
        rt.gpt_prefill(refer, [200, 150, 90], texts, 5, [1, 2, 3], **kw)
        assert rt.gpt_steps() == 1
        if mode == "eager":
            for _ in range(Gn - 1):
                rt.gpt_decode_step()
        elif mode == "graph":
            assert rt.gpt_decode(1000) == Gn - 1              # clamped to the session length
        else:
            assert rt.gpt_decode(17) == 17
            for _ in range(5):
                rt.gpt_decode_step()
            assert rt.gpt_decode(1000) == Gn - 1 - 22
        assert rt.gpt_steps() == Gn and not rt.gpt_all_finished()
        codes, n, lat = rt.gpt_finish()
        assert np.array_equal(codes, ref_codes) and np.array_equal(n, ref_n), mode
        assert torch.equal(lat, ref_lat), mode
    rt.set_option("gpt_graph", 0)
    codes, n, lat = rt.gpt_generate(refer, [200, 150, 90], texts, 5, [1, 2, 3], **kw)
    assert np.array_equal(codes, ref_codes) and torch.equal(lat, ref_lat)


def test_gpt_batch_above_one_session(rt, weights):
    """11 rows = two decode sessions (8 + 3): every row equals the row generated alone."""
    rs = np.random.RandomState(15)
    B = 11
    refer = dev((rs.randn(B, 128, 60) * 2 - 5).astype(np.float32))
    texts = [np.concatenate([rs.randint(3, 255, 4 + b), [0]]) for b in range(B)]
    codes, n, lat = rt.gpt_generate(refer, None, texts, 9, list(range(20, 20 + B)), max_generate_length=7, suppress_eos=True)
    for b in (0, 7, 8, 10):
        c1, _, l1 = rt.gpt_generate(refer[b:b + 1], None, [texts[b]], 9, [20 + b], max_generate_length=7, suppress_eos=True)
        assert np.array_equal(codes[b], c1[0]), b
        assert float((lat[b] - l1[0]).abs().max()) < 1e-5


@pytest.fixture(scope="module")
def model(weights):
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    return SynthesizerTrn(weights, folded=True)


def test_configs2_batch8_full_size_batch_invariance(model):
    """BASELINE configs[2] as the bench runs it (B = 8, 10 s prompts, 234 free-sampled codes -> T = 936): rows of the batch equal
    the rows generated alone (same seed / stream id), and a second run is bit-identical."""
    rs = np.random.RandomState(1)
    B = 8
    refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32))
    text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 60)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
    kw = dict(seed=1234, max_generate_length=235, suppress_eos=True, return_lengths=True)
    ids = list(range(B))
    wav, lens = model.infer(text, torch.full((B,), 61), refer, torch.full((B,), 936), batch=True, sample_ids=ids, **kw)
    assert lens == [234 * 1024] * B and torch.isfinite(wav).all()
    wav2, _ = model.infer(text, torch.full((B,), 61), refer, torch.full((B,), 936), batch=True, sample_ids=ids, **kw)
    assert torch.equal(wav, wav2)
    for b in (0, 5, 7):
        alone, _ = model.infer(text[b:b + 1], torch.tensor([61]), refer[b:b + 1], torch.tensor([936]), batch=True, sample_ids=[b], **kw)
        d = (wav[b] - alone[0]).double()
        tol(f"configs2_row{b}_vs_alone_rms", float(d.pow(2).mean().sqrt()), 1e-7)
        assert float(wav[b].double().pow(2).mean().sqrt()) > 1e-3


def test_configs4_long_form_60s_batch4(model):
    """BASELINE configs[4]: 60 s utterances (n = 1406 codes, T = 5624), batch 4 with ragged lengths, vocoder streamed: the streamed
    waveform equals the one-shot one, rows equal rows alone, nothing overflows at the long length."""
    rs = np.random.RandomState(78)
    B, n = 4, 1406
    ns = [1406, 1406, 1333, 1190]
    refer = torch.from_numpy((rs.randn(B, 128, 300) * 2 - 5).astype(np.float32))
    text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 30)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
    codes = [rs.randint(0, 8192, size=k) for k in ns]
    kw = dict(batch=True, seed=11, return_lengths=True)
    wav, lens = model.infer(text, torch.full((B,), 31), refer, torch.full((B,), 300), sample_ids=list(range(B)), forced_codes=codes, **kw)
    assert tuple(wav.shape) == (B, 1, n * 1024) and lens == [k * 1024 for k in ns]
    for b in range(B):
        w = wav[b, 0, :lens[b]]
        assert bool(torch.isfinite(w).all()) and float(w.pow(2).mean().sqrt()) > 1e-4
    ws, _ = model.infer(text, torch.full((B,), 31), refer, torch.full((B,), 300), sample_ids=list(range(B)), forced_codes=codes,
                        stream_vocoder=True, **kw)
    for b in range(B):
        tol(f"configs4_row{b}_streamed_vs_oneshot_maxabs", float((ws[b, 0, :lens[b]] - wav[b, 0, :lens[b]]).abs().max()), 1e-7)
    alone, _ = model.infer(text[3:4], torch.tensor([31]), refer[3:4], torch.tensor([300]), sample_ids=[3], forced_codes=codes[3:4], **kw)
    d = (wav[3, 0, :lens[3]] - alone[0, 0]).double()
    tol("configs4_row3_vs_alone_rms", float(d.pow(2).mean().sqrt()), 1e-7)


# ------------------------------------------------------------------------------------------------ device sampler
@pytest.mark.parametrize("top_k", [50, None])
def test_device_sampler_on_peaked_logits_fixture(rt, golden, top_k):
    """The reference's HF processors on peaked logits (tests/golden/sampler_filter.npz): the DEVICE sampler returns, for uniforms
    swept through the kept tokens' CDF (interval mid-points, near both edges, and a random sweep), exactly the token of the
    oracle's inverse-CDF draw; tokens the processors removed are never drawn."""
    from oracle import gpt as Gm
    g = golden("sampler_filter")
    rs = np.random.RandomState(3)
    for r in range(3):
        f = Gm.process_logits(g["scores"][r], g["history"][r], top_k=top_k)
        ref_f = g["filtered_k50" if top_k else "filtered_none"][r]
        assert np.array_equal(np.isfinite(f), np.isfinite(ref_f))
        p = np.exp((f - f[np.isfinite(f)].max()).astype(np.float64))
        p[~np.isfinite(f)] = 0.0
        c = np.cumsum(p) / p.sum()
        kept = np.nonzero(p > 0)[0]
        us, want = [], []
        lo = np.concatenate([[0.0], c[:-1]])
        for v in kept:
            w = c[v] - lo[v]
            if w < 2e-5:                       # narrower than the fp32 resolution of the device's own CDF: edge order undefined
                continue
            for frac in (0.5, 0.02, 0.98):
                us.append(lo[v] + frac * w)
                want.append(v)
        for u in rs.rand(64):
            v = int(min(np.searchsorted(c, u, side="right"), c.size - 1))
            if min(u - lo[v], c[v] - u) > 1e-5:
                us.append(u)
                want.append(v)
        us, want = np.array(us, np.float32), np.array(want)
        assert len(us) > 40
        got = []
        logits = dev(np.repeat(g["scores"][r:r + 1], 8, 0))
        hist = np.repeat(g["history"][r:r + 1], 8, 0)
        for i in range(0, len(us), 8):
            u8 = np.resize(us[i:i + 8], 8).astype(np.float32)
            got.extend(rt.op_sample_logits(logits, hist, dev(u8), top_k=top_k or 0).tolist()[:len(us[i:i + 8])])
        got = np.array(got)
        assert np.array_equal(got, want), (r, top_k, np.nonzero(got != want)[0][:5], got[got != want][:5], want[got != want][:5])
        assert set(got.tolist()) <= set(kept.tolist())


def test_sampler_steps_teacher_forced_golden(rt, golden):
    """Every p_sample of the reference's trace with x TEACHER-FORCED from the fixture: steps 49, 48, 47 (device Philox noise) and
    the last step i = 0 (no noise): x_after and pred_xstart per step (vqvae/utils/diffusion.py:445-485)."""
    g = golden("diff_sampler_steps")
    ce = dev(g["code_emb"])
    seed, sid = int(g["seed"]), [int(g["sample_id"])]
    x = g["x_init"]
    for i in (49, 48, 47):
        x1, x0 = rt.diff_p_sample(dev(x), ce, i, seed, sid, return_x0=True)
        e0, e1 = maxabs(host(x0), g[f"x0_after_{i}"]), maxabs(host(x1), g[f"x_after_{i}"])
        assert e0 < 5e-3 and e1 < 2e-3, (i, e0, e1)
        x = g[f"x_after_{i}"]                                    # teacher forcing: the reference's own x
    x1 = rt.diff_p_sample(dev(g["x_before_0"]), ce, 0, seed, sid)
    assert maxabs(host(x1), g["x_after_0"]) < 2e-4, maxabs(host(x1), g["x_after_0"])


# ----------------------------------------------------------------------------------------------------------------------------
# Headline-size END-TO-END parity: the reference's own SynthesizerTrn.infer at 234 codes / T_ref = 936 (e2e_fullsize.npz)
# ----------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def E(golden):
    return golden("e2e_fullsize")


@pytest.fixture(scope="module")
def EI():
    from fullsize_inputs import e2e_inputs
    return e2e_inputs()


@pytest.fixture(scope="module")
def synth(weights):
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    return SynthesizerTrn(weights, folded=True)


def _ragged_batch8(EI, row=5):
    """The fixture's utterance as row `row` of a ragged batch of 8: other rows have other prompts / texts / code counts."""
    rs = np.random.RandomState(91)
    B = 8
    n = [234, 180, 201, 97, 234, N_CODES, 150, 222]
    rl = [936, 700, 936, 512, 801, T, 936, 640]
    tl = [61, 40, 61, 25, 50, 61, 61, 33]
    refer = (rs.randn(B, 128, T) * 2 - 5).astype(np.float32)
    text = np.zeros((B, 61), np.int32)
    for b in range(B):
        text[b, : tl[b] - 1] = rs.randint(3, 255, tl[b] - 1)
    codes = [rs.randint(0, 8192, size=n[b]) for b in range(B)]
    refer[row], text[row], codes[row] = EI["refer"][0], EI["text"][0], EI["codes"][0]
    assert n[row] == N_CODES and rl[row] == T and tl[row] == 61
    return refer, rl, text, tl, codes, n


@pytest.mark.parametrize("x3", [1, 0])
def test_e2e_234_codes_B1_vs_reference_waveform(synth, E, EI, x3):
    """north_star: 'outputs match the reference CPU path sample-for-sample within 1e-3 RMS on the 24 kHz waveform' - at the
    configuration the headline number is quoted on: 10 s prompt, 234 codes, 50 sampling steps, both kernel sets, B = 1."""
    synth.rt.set_option("conv_x3", x3)
    try:
        wav = synth.infer(torch.from_numpy(EI["text"]), torch.tensor([61]), torch.from_numpy(EI["refer"]), torch.tensor([T]),
                          seed=int(E["seed"]), sample_ids=[int(E["sample_id"])], forced_codes=[EI["codes"][0]])
    finally:
        synth.rt.set_option("conv_x3", 1)
    w = host(wav)[0, 0]
    assert w.shape == E["wav"].shape
    r = rms(w, E["wav"])
    print(f"\n[e2e 234 codes, B=1, conv_x3={x3}] waveform RMS error {r:.3e} (reference RMS {float(E['wav_rms']):.3e}), max-abs {maxabs(w, E['wav']):.3e}")
    # north_star asks for 1e-3 RMS; with the seed-0 weights generator(z = 0) alone is within 1.04e-3 of the reference, so that limit
    # would prove nothing: the gate sits at ~20 x the measured error (4.3e-9), 4 orders below the z-driven part of the signal
    tol(f"e2e_B1_wav_rms_x3={x3}", r, 1e-7)


@pytest.mark.parametrize("x3", [1, 0])
def test_e2e_234_codes_row5_of_ragged_B8_vs_reference_waveform(synth, E, EI, x3):
    """The same utterance as row 5 of a RAGGED batch of 8 (configs[2]'s batch; other rows shorter prompts / texts / code counts):
    the batch-8 kernels (two CFG stream chunks, 240-tile launches, 8-row decode session shapes) against the reference's waveform."""
    refer, rl, text, tl, codes, n = _ragged_batch8(EI)
    synth.rt.set_option("conv_x3", x3)
    try:
        wav, lens = synth.infer(torch.from_numpy(text), torch.tensor(tl), torch.from_numpy(refer), torch.tensor(rl), batch=True,
                                seed=int(E["seed"]), sample_ids=[100, 101, 102, 103, 104, int(E["sample_id"]), 106, 107],
                                forced_codes=codes, return_lengths=True)
    finally:
        synth.rt.set_option("conv_x3", 1)
    assert lens == [1024 * v for v in n]
    w = host(wav)[5, 0, : lens[5]]
    r = rms(w, E["wav"])
    print(f"\n[e2e 234 codes, row 5 of ragged B=8, conv_x3={x3}] waveform RMS error {r:.3e}, max-abs {maxabs(w, E['wav']):.3e}")
    tol(f"e2e_row5_of_B8_wav_rms_x3={x3}", r, 1e-7)


def test_e2e_two_rows_of_a_ragged_B8_batch_both_vs_reference_waveforms(synth, E, EI, golden):
    """A ragged batch of 8 in which TWO rows are pinned by the reference's own waveforms: row 5 = the 234-code / 936-frame utterance
    (e2e_fullsize.npz), row 2 = a SHORTER one (150 codes, 700-frame prompt, 40 text ids: e2e_fullsize_b.npz) - padded prompt, padded
    text, padded codes, a different sequence length in every kernel of stages A, B and C - each against the run the reference made of
    that utterance ALONE (vqvae/model_24k.py:774-810 is batch 1)."""
    from fullsize_inputs import e2e_inputs_b
    EB, J = golden("e2e_fullsize_b"), e2e_inputs_b()
    refer, rl, text, tl, codes, n = _ragged_batch8(EI)
    refer[2] = 0.0
    refer[2, :, :700] = J["refer"][0]
    rl[2] = 700
    text[2] = 0
    text[2, :41] = J["text"][0]
    tl[2] = 41
    codes[2] = J["codes"][0]
    n[2] = 150
    sids = [100, 101, int(EB["sample_id"]), 103, 104, int(E["sample_id"]), 106, 107]
    wav, lens = synth.infer(torch.from_numpy(text), torch.tensor(tl), torch.from_numpy(refer), torch.tensor(rl), batch=True,
                            seed=int(E["seed"]), sample_ids=sids, forced_codes=codes, return_lengths=True)
    assert lens == [1024 * v for v in n]
    for row, ref in ((5, E), (2, EB)):
        w = host(wav)[row, 0, : lens[row]]
        r = rms(w, ref["wav"])
        print(f"\n[ragged B=8, row {row}: {lens[row] // 1024} codes] waveform RMS error {r:.3e} (reference RMS {float(ref['wav_rms']):.3e})")
        tol(f"e2e_two_rows_row{row}_wav_rms", r, 1e-7)
        assert np.all(host(wav)[row, 0, lens[row]:] == 0)


@pytest.mark.parametrize("x3", [1, 0])
def test_e2e_sampler_drift_along_the_50_step_chain(synth, E, EI, x3):
    """Where along the chain does the error grow?  The sampler state after p_sample 49 / 40 / 25 / 0 and the de-normalised mel vs the
    reference's own trace (the eps amplification sqrt(1/abar - 1) is 153x at the first step).  Tolerances are per checkpoint;
    the measured errors are printed so that drift is visible in the log."""
    rt = synth.rt
    rt.set_option("conv_x3", x3)
    try:
        refer = dev(EI["refer"])
        lat = rt.gpt_latents(refer, [T], [EI["text"][0]], [EI["codes"][0]])
        code_emb = rt.diff_timestep_independent(lat, rt.diff_conditioning(refer, [T]), [N_CODES])
        errs = {}
        for i in (49, 40, 25, 0):
            x = host(rt.diff_sample(code_emb, int(E["seed"]), [int(E["sample_id"])], lens=[T], n_steps=50 - i, denorm=False))[0]
            s, t = sub(x, E)
            errs[i] = max(maxabs(s, E[f"x_after_{i}_s"]), maxabs(t, E[f"x_after_{i}_t"]))
        mel = host(rt.diff_sample(code_emb, int(E["seed"]), [int(E["sample_id"])], lens=[T], denorm=True))[0]
        s, t = sub(mel, E)
        errs["mel"] = max(maxabs(s, E["mel_s"]), maxabs(t, E["mel_t"]))
    finally:
        rt.set_option("conv_x3", 1)
    print(f"\n[sampler drift, conv_x3={x3}] max-abs vs the reference after step 49/40/25/0 and on the de-normalised mel: " +
          ", ".join(f"{k}: {v:.2e}" for k, v in errs.items()))
    # limits ~ 20 x the measured drift (r03: 7e-7 / 1.2e-6 / 4.5e-6 / 1.4e-5, mel 1.0e-4 split precision, 1.6e-4 exact fp32)
    for k, lim in ((49, 2e-5), (40, 4e-5), (25, 1e-4), (0, 2e-4), ("mel", 2e-3)):        # de-normalisation scales by (2.7 + 11.51) / 2 = 7.1
        tol(f"drift_x3={x3}_{k}", errs[k], lim)


def test_free_sampling_234_tokens_vs_reference_hf_loop(golden, EI, weights):
    """Token identity with the reference's own UNCACHED HF sampling loop (gpt/model.py:514-545) over 234 tokens at the 936-frame
    prompt - where a KV-cache summation-order drift could flip a draw.  On a mismatch the first divergent step and the reference's
    CDF margin there are reported: a margin < 1e-6 is an acceptable tie (the draw sits on a CDF edge), anything else a bug."""
    from detail_tts_amd.runtime import Runtime
    g = golden("gpt_generate_fullsize")
    rt = Runtime(weights, folded=True, parts=("gpt",))
    ref = g["codes"][0]
    for B, wgs in ((1, 0), (8, 0), (8, 64), (8, 32)):      # alone; as row 3 of an 8-row decode session (the bench's session shape) on the 128-workgroup
        rs = np.random.RandomState(17)                      # token kernel and - round 6 - on the 64- / 32-workgroup ones (dtts_gpt_options.token_wgs)
        refer = (rs.randn(B, 128, T) * 2 - 5).astype(np.float32)
        texts = [np.concatenate([rs.randint(3, 255, 60), [0]]).astype(np.int32) for _ in range(B)]
        sids = [200 + b for b in range(B)]
        row = 0 if B == 1 else 3
        refer[row], texts[row], sids[row] = EI["refer"][0], EI["text"][0], int(g["sample_id"])
        codes, ncodes, _ = rt.gpt_generate(dev(refer), [T] * B, texts, int(g["seed"]), sids, max_generate_length=N_CODES + 1, suppress_eos=True,
                                           token_wgs=wgs)
        got = np.asarray(codes[row][: ref.size])
        if not np.array_equal(got, ref):
            k = int(np.nonzero(got != ref)[0][0])
            margin = float(g["f64_margins"][k])
            assert margin < 1e-6, f"B={B} wgs={wgs}: first divergent step {k}: got {got[k]}, reference {ref[k]}, CDF margin {margin:.3e} (not a tie)"
            print(f"\n[free sampling, B={B}] tie at step {k} (CDF margin {margin:.3e}); identical before it")
        else:
            print(f"\n[free sampling, B={B}] all {ref.size} tokens identical to the reference's HF loop (min CDF margin {float(g['f64_margins'].min()):.2e})")
