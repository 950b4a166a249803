"""GPU parity: HIP stage C (MelStyleEncoder, enc_p, inverse flow, HiFiGAN generator) vs oracle + golden."""
import numpy as np
import pytest

from conftest import tol

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
def rt(weights):
    from detail_tts_amd.runtime import Runtime
    return Runtime(weights, folded=True, parts=("vocoder",))


def test_ref_enc_golden(rt, weights, golden):
    """Row 0 (unpadded) is pinned by the reference fixture.  Row 1 is padded (29 of 40 frames): the reference's
    padded-batch arithmetic leaks Mish(bias) of the padding frames through the k=5 convs into the last valid frames,
    which never happens on its real path (infer is batch 1, unpadded).  A batch here is DEFINED as independent
    single-utterance runs, so row 1 is checked against the oracle run on the truncated utterance."""
    from oracle import gpt as G
    g = golden("mel_style")
    out = host(rt.mel_style("ref_enc", dev(g["x2"]), g["len2"]))
    assert maxabs(out[0], g["ref_enc2"][0, :, 0]) < 1e-4
    L = int(g["len2"][1])
    alone = G.mel_style_encoder(weights, "ref_enc", g["x2"][1:2, :, :L], [L])[0, :, 0]
    assert maxabs(out[1], alone) < 1e-4


def test_generator_golden(rt, golden):
    g = golden("vocoder")
    wav = host(rt.generator(dev(g["z"]), dev(g["g"][:, :, 0])))
    assert wav.shape == g["wav"].shape
    tol("generator_golden_maxabs", maxabs(wav, g["wav"]), 5e-7)           # ~20 x measured; the seed-0 waveform is 7e-3 RMS
    assert float(np.abs(g["wav"]).max()) > 1e-3     # the fixture is not a trivially small signal


def test_infer_flowvae_golden(rt, golden):
    g = golden("vocoder")
    wav, z = rt.vocoder(dev(g["mel"]), int(g["seed"]), [int(g["sample_id"])], return_z=True)
    assert maxabs(host(z), g["z"]) < 2e-4, maxabs(host(z), g["z"])
    tol("infer_flowvae_golden_wav_rms", rms(host(wav), g["wav"]), 1e-7)   # see tests/test_gpu_signal.py for the z-driven weight set


def test_enc_p_unit_entry_vs_reference_and_oracle(rt, golden, weights):
    """dtts_op_enc_p = in_proj + SpecEncoder (vqvae/model_24k.py:856-857 -> :71-107, attentions.py:73-107,161-303): m_p / logs_p vs the
    REFERENCE's own (vocoder.npz), and a ragged batch of 2 vs the oracle per row."""
    from oracle import ops, vocoder as V
    g = golden("vocoder")
    m_p, logs_p = rt.op_enc_p(dev(g["mel"]))
    assert maxabs(host(m_p), g["m_p"]) < 2e-4 and maxabs(host(logs_p), g["logs_p"]) < 2e-4, (maxabs(host(m_p), g["m_p"]), maxabs(host(logs_p), g["logs_p"]))
    assert float(np.abs(g["m_p"]).max()) > 1e-2
    rs = np.random.RandomState(12)
    mel = (rs.randn(2, 128, 44) * 2 - 5).astype(np.float32)
    lens = [44, 31]
    m_p, logs_p = rt.op_enc_p(dev(mel), lens=lens)
    for b, L in enumerate(lens):
        x = ops.conv1d(mel[b:b + 1, :, :L], weights["in_proj.weight"], weights["in_proj.bias"], padding=1)
        _, rm, rl = V.spec_encoder(weights, x, [L])
        assert maxabs(host(m_p)[b, :, :L], rm[0]) < 2e-4 and maxabs(host(logs_p)[b, :, :L], rl[0]) < 2e-4


def test_vocoder_varlen_batch_vs_oracle(rt, weights):
    from oracle import vocoder as V
    rs = np.random.RandomState(11)
    mel = (rs.randn(2, 128, 40) * 2 - 5).astype(np.float32)
    lens = [40, 28]
    wav = host(rt.vocoder(dev(mel), 77, [4, 5], lens=lens))
    for b, L in enumerate(lens):
        ref = V.infer_flowvae(weights, mel[b:b + 1, :, :L], [L], 77, [4 + b])[0, 0]
        got = wav[b, 0, :256 * L]
        tol(f"vocoder_varlen_row{b}_vs_oracle_rms", rms(got, ref), 1e-7)
        assert np.all(wav[b, 0, 256 * L:] == 0)


def test_vocoder_rejects_bad_length(rt):
    from detail_tts_amd.runtime import DttsError
    with pytest.raises(DttsError):
        rt.vocoder(torch.zeros(1, 128, 42, device="cuda"), 0, [0])              # T % 4 != 0 (model_24k.py:851)


def test_generator_long_input_halo_consistency(rt, weights):
    """Chunk invariance: the generator is purely convolutional, so the middle of a long input equals the same frames
    synthesised inside a shorter window with a 16-frame halo (SURVEY App. B receptive field)."""
    rs = np.random.RandomState(12)
    z = rs.randn(1, 192, 160).astype(np.float32)
    g = rs.randn(1, 768).astype(np.float32) * 0.1
    full = host(rt.generator(dev(z), dev(g)))[0, 0]
    part = host(rt.generator(dev(z[:, :, 40:120]), dev(g)))[0, 0]
    a, b = full[(40 + 16) * 256:(120 - 16) * 256], part[16 * 256:(80 - 16) * 256]
    tol("generator_halo_consistency_maxabs", maxabs(a, b), 1e-7)


# ---------------------------------------------------------------------------- infer_gpt's VQ decode path
@pytest.fixture(scope="module")
def rt_vq(weights):
    from detail_tts_amd.runtime import Runtime
    return Runtime(weights, folded=True, parts=("vocoder", "vq"))


def test_vq_decode_golden(rt_vq, golden):
    g = golden("vq_path")
    mel = host(rt_vq.vq_decode([g["codes"][0]], dev(g["refer"])))
    assert mel.shape == g["recon"].shape
    assert maxabs(mel, g["recon"]) < 2e-4, maxabs(mel, g["recon"])
    wav = host(rt_vq.vocoder(dev(mel), int(g["seed"]), [int(g["sample_id"])]))
    tol("vq_decode_wav_rms", rms(wav, g["wav"]), 1e-7)


def test_vq_decode_varlen_batch_vs_oracle(rt_vq, weights):
    from oracle import vq
    rs = np.random.RandomState(21)
    refer = (rs.randn(2, 128, 50) * 2 - 5).astype(np.float32)
    rl = [50, 37]
    codes = [rs.randint(0, 8192, size=23), rs.randint(0, 8192, size=14)]
    mel = host(rt_vq.vq_decode(codes, dev(refer), rl))
    assert mel.shape == (2, 128, 92)
    for b in range(2):
        ref = vq.vq_decode_mel(weights, codes[b][None], refer[b:b + 1, :, :rl[b]], [rl[b]])[0]
        assert maxabs(mel[b, :, :4 * len(codes[b])], ref) < 2e-4, b
        assert np.all(mel[b, :, 4 * len(codes[b]):] == 0)


def test_vq_decode_rejects_out_of_codebook(rt_vq):
    from detail_tts_amd.runtime import DttsError
    with pytest.raises(DttsError):
        rt_vq.vq_decode([np.array([5, 8192])], torch.zeros(1, 128, 20, device="cuda"))      # start/stop tokens are not codebook rows


def test_vq_encode_golden_and_diverse_codebook(weights, golden):
    """encode: the reference fixture (seed-0 weights send every frame to one code), then a codebook re-centred on the projected
    frames so that the nearest-entry search is exercised with many different winners (bit-exact indices vs the oracle)."""
    from detail_tts_amd.runtime import Runtime
    from oracle import vq
    g = golden("vq_encode")
    r = Runtime(weights, folded=True, parts=("vocoder", "vq"))
    codes, xvq = r.vq_encode(dev(g["mel"]))
    assert maxabs(host(xvq), g["x_vq"]) < 1e-4
    assert np.array_equal(host(codes).astype(np.int64), g["codes"])

    rs = np.random.RandomState(31)
    mel = (rs.randn(2, 128, 120) * 2 - 5).astype(np.float32)
    lens = [120, 90]
    P = dict(weights)
    x8, _ = vq.quantize_distances(P, vq.vq_enc(P, mel))
    emb = P["quantizer.vq.layers.0._codebook.embed"].copy()
    pts = x8.reshape(-1, 8)
    emb[:] = pts[rs.randint(0, len(pts), len(emb))] + rs.randn(*emb.shape).astype(np.float32) * float(pts.std()) * 0.5
    P["quantizer.vq.layers.0._codebook.embed"] = emb
    r2 = Runtime(P, folded=True, parts=("vocoder", "vq"))
    codes, xvq = r2.vq_encode(dev(mel), lens)
    codes = host(codes)
    for b, L in enumerate(lens):
        ref_codes, ref_x = vq.encode(P, mel[b:b + 1, :, :L])
        n = ref_codes.shape[1]
        assert n == (L + 3) // 4
        assert maxabs(host(xvq)[b, :, :n], ref_x[0]) < 1e-4
        same = codes[b, :n] == ref_codes[0]
        if not same.all():          # a flipped index is only acceptable on an fp32-level tie of the two distances
            _, d = vq.quantize_distances(P, ref_x)
            for t in np.nonzero(~same)[0]:
                assert abs(d[0, t, codes[b, t]] - d[0, t, ref_codes[0, t]]) < 1e-5 * max(1.0, abs(d[0, t, ref_codes[0, t]]))
        assert len(np.unique(ref_codes)) > 8


def test_infer_vqvae_roundtrip_surface(weights):
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    from oracle import vq
    m = SynthesizerTrn(weights, folded=True)
    rs = np.random.RandomState(32)
    mel = (rs.randn(1, 128, 40) * 2 - 5).astype(np.float32)
    codes, x_vq = m.encode(torch.from_numpy(mel), torch.tensor([40]))
    ref_codes, _ = vq.encode(weights, mel)
    assert codes.dtype == torch.int64 and np.array_equal(codes.cpu().numpy(), ref_codes)
    recon, wav = m.infer_vqvae(torch.from_numpy(mel), seed=3, sample_ids=[9])
    ref_mel = vq.vq_decode_mel(weights, ref_codes, mel, [40])
    assert maxabs(recon.cpu().numpy(), ref_mel) < 2e-4
    assert tuple(wav.shape) == (1, 1, 40 * 256) and bool(torch.isfinite(wav).all())


def test_generator_stream_equals_full_and_wav_writer(weights, tmp_path):
    """chunked generator with halo == one-shot generator; the WAV writer round-trips through the stdlib reader."""
    import wave
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn, write_wav
    m = SynthesizerTrn(weights, folded=True)
    rs = np.random.RandomState(41)
    z = torch.from_numpy(rs.randn(1, 192, 150).astype(np.float32)).cuda()
    g = torch.from_numpy((rs.randn(1, 768, 1) * 0.1).astype(np.float32)).cuda()
    full = m.dec(z, g=g)
    parts = list(m.dec.stream(z, g, chunk=48))
    assert [p.shape[-1] for p in parts] == [48 * 256, 48 * 256, 48 * 256, 6 * 256]
    cat = torch.cat(parts, -1)
    assert cat.shape == full.shape
    tol("generator_stream_vs_full_maxabs", float((cat - full).abs().max()), 1e-7)
    path = tmp_path / "gen.wav"
    write_wav(path, full[0], 24000)
    with wave.open(str(path), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 24000, 150 * 256)
        pcm = np.frombuffer(f.readframes(f.getnframes()), np.int16)
    assert np.abs(pcm / 32767.0 - full[0, 0].clamp(-1, 1).cpu().numpy()).max() < 1.0 / 32767 + 1e-6


@pytest.mark.parametrize("stage,branch", [(0, 0), (0, 2), (1, 1), (2, 0), (2, 1), (2, 2), (3, 0), (3, 1), (3, 2), (4, 0), (4, 1), (4, 2)])
def test_unit_resblock1_vs_oracle(rt, weights, stage, branch):
    """dtts_op_resblock1: one HiFiGAN ResBlock1 (kernel 3 / 7 / 11, dilations 1, 3, 5) against the oracle, ragged batch."""
    from oracle import vocoder as V
    rs = np.random.RandomState(20 + stage * 3 + branch)
    ch = 400 >> (stage + 1)
    T = 160
    x = (rs.randn(2, ch, T) * 0.5).astype(np.float32)
    lens = [T, T - 37]
    y = host(rt.op_resblock1(stage, branch, dev(x), lens))
    for b, L in enumerate(lens):
        ref = V.resblock1(weights, f"dec.resblocks.{stage * 3 + branch}", x[b:b + 1, :, :L], (3, 7, 11)[branch])[0]
        assert maxabs(y[b, :, :L], ref) < 2e-5 * max(1.0, float(np.abs(ref).max())), (stage, branch, b)


@pytest.mark.parametrize("stage,branch", [(3, 0), (3, 2), (4, 1), (4, 2)])
def test_fused_resblock1_across_time_tiles_vs_oracle(rt, weights, stage, branch):
    """The LDS-resident fused ResBlock1 of the narrow generator stages (resblock1_fused.hip: 25 and 12 channels) on sequences that span
    several time tiles (interiors of 392 / 968 samples + 60-sample halos): tile seams, a ragged row ending inside a tile, and a row
    ending inside a halo - every element against the oracle."""
    from oracle import vocoder as V
    rs = np.random.RandomState(60 + stage * 3 + branch)
    ch = 400 >> (stage + 1)
    T = 2600
    x = (rs.randn(3, ch, T) * 0.5).astype(np.float32)
    lens = [T, 1000, 1990]
    y = host(rt.op_resblock1(stage, branch, dev(x), lens))
    for b, L in enumerate(lens):
        ref = V.resblock1(weights, f"dec.resblocks.{stage * 3 + branch}", x[b:b + 1, :, :L], (3, 7, 11)[branch])[0]
        assert maxabs(y[b, :, :L], ref) < 2e-5 * max(1.0, float(np.abs(ref).max())), (stage, branch, b, maxabs(y[b, :, :L], ref))


def test_split_precision_range_check_catches_saturation(weights):
    """Option x3_range_check: an activation beyond the fp16 planes' range (|x| > 65504 / 16) in a wide ResBlock1 fails the call with a
    message naming the exact-fp32 switch, instead of a silently saturated conv; in-range inputs pass and equal the unchecked result."""
    from detail_tts_amd.runtime import DttsError, Runtime
    rt2 = Runtime(weights, folded=True, parts=("vocoder",))
    rs = np.random.RandomState(70)
    x = (rs.randn(1, 200, 160) * 0.5).astype(np.float32)
    base = host(rt2.op_resblock1(0, 1, dev(x)))
    rt2.set_option("x3_range_check", 1)
    assert np.array_equal(host(rt2.op_resblock1(0, 1, dev(x))), base)
    x[0, 17, 40] = 5000.0
    with pytest.raises(DttsError, match="conv_x3"):
        rt2.op_resblock1(0, 1, dev(x))
    rt2.set_option("conv_x3", 0)                       # the exact fp32 kernels take any finite input
    assert np.isfinite(host(rt2.op_resblock1(0, 1, dev(x)))).all()
    rt2.set_option("conv_x3", 1)                       # the fused narrow-stage kernel splits inside the workgroup: same check
    y = (rs.randn(1, 12, 300) * 0.5).astype(np.float32)
    host(rt2.op_resblock1(4, 2, dev(y)))
    y[0, 3, 150] = -60000.0                            # lrelu -> -6000: beyond +-4094
    with pytest.raises(DttsError, match="conv_x3"):
        rt2.op_resblock1(4, 2, dev(y))


@pytest.mark.parametrize("flow", [0, 3])
def test_unit_wn_vs_oracle(rt, weights, flow):
    """dtts_op_wn: the WaveNet of one residual coupling layer (gated tanh * sigmoid, res / skip convs, global conditioning)."""
    from oracle import vocoder as V
    rs = np.random.RandomState(40 + flow)
    T = 120
    h = (rs.randn(2, 192, T) * 0.7).astype(np.float32)
    g = (rs.randn(2, 768) * 0.3).astype(np.float32)
    lens = [T, T - 29]
    out = host(rt.op_wn(flow, dev(h), dev(g), lens))
    for b, L in enumerate(lens):
        mf = np.ones((1, 1, L), np.float32)
        ref = V.wn(weights, f"flow.flows.{2 * flow}.enc", h[b:b + 1, :, :L], mf, g[b:b + 1, :, None])[0]
        assert maxabs(out[b, :, :L], ref) < 2e-5 * max(1.0, float(np.abs(ref).max())), (flow, b)
        assert np.all(out[b, :, L:] == 0)


def test_range_check_is_on_by_default_and_fails_the_request_that_saturated(weights):
    """Without any option the kernels still raise a (host-mapped) saturation flag; no synchronisation is added.  Every stage-C call
    takes a ticket, and dtts_vocoder_check(ticket) - called once the caller has waited for that call's output, as SynthesizerTrn.infer /
    infer_stream do - fails THE REQUEST THAT SATURATED: not the innocent next call on the handle, and also the last call of a stream
    (ADVICE r03 / r04)."""
    from detail_tts_amd.runtime import DttsError, Runtime
    rt2 = Runtime(weights, folded=True, parts=("vocoder",))
    rs = np.random.RandomState(71)
    x = (rs.randn(1, 200, 160) * 0.5).astype(np.float32)
    ok = host(rt2.op_resblock1(0, 1, dev(x)))
    t_ok = rt2.vocoder_ticket()
    bad = x.copy()
    bad[0, 17, 40] = 5000.0
    rt2.op_resblock1(0, 1, dev(bad))                   # saturates; returns (nothing synchronises on the flag here)
    t_bad = rt2.vocoder_ticket()
    assert t_bad == t_ok + 1
    after = rt2.op_resblock1(0, 1, dev(x))             # the NEXT call is not aborted ...
    t_after = rt2.vocoder_ticket()
    torch.cuda.synchronize()
    assert np.array_equal(host(after), ok)             # ... and is unaffected
    rt2.vocoder_check(t_ok)
    rt2.vocoder_check(t_after)
    with pytest.raises(DttsError, match="conv_x3"):
        rt2.vocoder_check(t_bad)                       # the request that saturated fails, also when it is not the last one issued
    rt2.vocoder_check(t_bad)                           # reported once
    # a flag nobody checks is recycled (reported on stderr) when its slot comes round again: 8 calls later nothing raises
    rt2.op_resblock1(0, 1, dev(bad))
    for _ in range(9):
        rt2.op_resblock1(0, 1, dev(x))
    torch.cuda.synchronize()
    rt2.vocoder_check(rt2.vocoder_ticket())


@pytest.mark.parametrize("stage,branch", [(0, 0), (0, 2), (1, 1)])
def test_resblock1_planes_chained_through_the_conv_epilogues_are_bit_identical(weights, stage, branch):
    """Round 5: in the wide generator stages every split-precision conv of a ResBlock1 writes leaky_relu(y) as the next conv's operand
    planes from its epilogue (option voc_chain_planes, default on) instead of storing fp32 and running a split pass: the same values
    split the same way, so the block's output must not change by one bit - ragged lengths inside / on / past a 192-column tile, channel
    counts that are not multiples of 8 (200, 100: zero chunk rows), and a saturating activation still fails its own ticket."""
    from detail_tts_amd.runtime import DttsError, Runtime
    rt2 = Runtime(weights, folded=True, parts=("vocoder",))
    rs = np.random.RandomState(80 + stage)
    ch = [200, 100][stage]
    T = 600
    lens = [600, 192, 385, 77]
    x = (rs.randn(4, ch, T) * 0.5).astype(np.float32)
    outs = {}
    try:
        for flag in (0, 1):
            rt2.set_option("voc_chain_planes", flag)
            outs[flag] = host(rt2.op_resblock1(stage, branch, dev(x), lens))
    finally:
        rt2.set_option("voc_chain_planes", 1)
    for b, L in enumerate(lens):
        assert float(np.abs(outs[1][b, :, :L]).max()) > 0.1
        assert np.array_equal(outs[0][b, :, :L], outs[1][b, :, :L]), b
    bad = x.copy()
    bad[1, 7, 100] = 5000.0
    rt2.op_resblock1(stage, branch, dev(bad), lens)
    t_bad = rt2.vocoder_ticket()
    torch.cuda.synchronize()
    with pytest.raises(DttsError, match="conv_x3"):
        rt2.vocoder_check(t_bad)


def test_wavenet_in_layers_on_the_split_precision_kernel_vs_fp32_path_and_range_ticket(weights):
    """Round 5: the flow's gated k = 5 WaveNet in_layers run as 1x1 conv_x3 launches over the 5-tap expansion of h
    (same w3 image, tanh * sigmoid + the conditioning rows in the epilogue).  Against the exact fp32-MFMA form (conv_x3 = 0) on ragged
    lengths that end inside / on / just past a 192-column tile, and: an activation beyond the fp16 planes' range raises THIS call's
    range-check ticket (the flow runs before the generator inside one vocoder call: one ticket per top-level call)."""
    from detail_tts_amd.runtime import DttsError, Runtime
    rt2 = Runtime(weights, folded=True, parts=("vocoder",))
    rs = np.random.RandomState(73)
    T = 388
    lens = [388, 192, 196, 40]
    h = (rs.randn(4, 192, T) * 0.7).astype(np.float32)
    g = (rs.randn(4, 768) * 0.3).astype(np.float32)
    outs = {}
    try:
        for flag in (1, 0):
            rt2.set_option("conv_x3", flag)
            outs[flag] = host(rt2.op_wn(1, dev(h), dev(g), lens))
    finally:
        rt2.set_option("conv_x3", 1)
    for b, L in enumerate(lens):
        a, r = outs[1][b, :, :L], outs[0][b, :, :L]
        assert float(np.abs(r).max()) > 0.05
        assert maxabs(a, r) < 1e-5 * max(1.0, float(np.abs(r).max())), (b, maxabs(a, r))
        assert np.all(outs[1][b, :, L:] == 0)
    t_ok = rt2.vocoder_ticket()
    bad = h.copy()
    bad[2, 50, 100] = 5000.0
    rt2.op_wn(1, dev(bad), dev(g), lens)
    t_bad = rt2.vocoder_ticket()
    assert t_bad == t_ok + 1
    rt2.op_wn(1, dev(h), dev(g), lens)
    torch.cuda.synchronize()
    rt2.vocoder_check(t_ok)
    rt2.vocoder_check(rt2.vocoder_ticket())
    with pytest.raises(DttsError, match="conv_x3"):
        rt2.vocoder_check(t_bad)
    # a streamed vocoder call (several generator windows) is ONE ticket
    mel = dev((rs.randn(1, 128, 96) * 2 - 5).astype(np.float32))
    t0 = rt2.vocoder_ticket()
    rt2.vocoder(mel, 3, [0], stream_chunk=32)
    assert rt2.vocoder_ticket() == t0 + 1
    rt2.vocoder(mel, 3, [0])
    assert rt2.vocoder_ticket() == t0 + 2


def test_calls_from_many_short_lived_host_threads(rt):
    """Every host thread that calls in gets its own pinned upload ring; rings of threads that are gone are recycled (at most 16 per
    handle) instead of accumulating.  24 threads, one after the other, each with lengths to upload: same result every time."""
    import threading
    rs = np.random.RandomState(72)
    mel = dev((rs.randn(2, 128, 40) * 2 - 5).astype(np.float32))
    ref = host(rt.mel_style("ref_enc", mel, [40, 31]))
    outs = []

    def work():
        torch.cuda.set_device(0)
        outs.append(host(rt.mel_style("ref_enc", mel, [40, 31])))

    for _ in range(24):
        t = threading.Thread(target=work)
        t.start()
        t.join()
    assert len(outs) == 24 and all(np.array_equal(o, ref) for o in outs)


def test_calls_from_more_than_sixteen_long_lived_host_threads(rt):
    """a pool of 20 LIVE worker threads that take turns on one handle (serialised: the "two at a time" contract holds) - the
    17th live thread used to fail permanently; now it gets an upload ring of its own.  Same result from every thread, twice round."""
    import threading
    rs = np.random.RandomState(73)
    mel = dev((rs.randn(2, 128, 40) * 2 - 5).astype(np.float32))
    ref = host(rt.mel_style("ref_enc", mel, [40, 29]))
    n = 20
    turn = [threading.Semaphore(0) for _ in range(n)]
    done = threading.Semaphore(0)
    outs, errs = [], []

    def work(i):
        torch.cuda.set_device(0)
        for _ in range(2):
            turn[i].acquire()
            try:
                outs.append(host(rt.mel_style("ref_enc", mel, [40, 29])))
            except Exception as e:          # noqa: BLE001
                errs.append(repr(e))
            done.release()

    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in ths:
        t.start()
    for _ in range(2):
        for i in range(n):
            turn[i].release()
            done.acquire()
    for t in ths:
        t.join()
    assert not errs, errs[:2]
    assert len(outs) == 2 * n and all(np.array_equal(o, ref) for o in outs)
