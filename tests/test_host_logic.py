"""CPU tests: host logic, packer layouts (checked against the oracle's conv arithmetic), C-ABI exports, N>1 sharding."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_and_hparams():
    from detail_tts_amd.config import HParams, load_config
    cfg = load_config({"diffusion": {"g_channels": 768, "num_layers": 10}})
    assert "g_channels" not in cfg["diffusion"] and cfg["diffusion"]["model_channels"] == 768
    h = HParams(**cfg)
    assert h.data.hop_length == 256 and h["vaegan"]["upsample_rates"] == [8, 4, 2, 2, 2] and "gpt" in h


def test_reference_config_file_matches_defaults():
    ref = "/root/reference/vqvae/configs/config_24k.json"
    if not os.path.exists(ref):
        pytest.skip("reference not present (GPU box)")
    from detail_tts_amd.config import DEFAULT_CONFIG, load_config
    cfg = load_config(ref)
    for blk in ("diffusion", "gpt"):
        for k, v in DEFAULT_CONFIG[blk].items():
            assert cfg[blk][k] == v, (blk, k)
    for k, v in DEFAULT_CONFIG["vaegan"].items():
        assert cfg["vaegan"][k] == v, k


def test_weight_spec_counts():
    from detail_tts_amd.weights import folded_param_names, inference_param_spec
    spec = inference_param_spec()
    n = sum(int(np.prod(s)) for s, _ in spec.values())
    assert abs(n - 270.946e6) < 1e4           # 266.35 M for infer + 4.59 M for the VQ decode / encode paths
    n_vq = sum(int(np.prod(s)) for k, (s, _) in spec.items() if k.startswith(("quantizer.", "vq_dec.", "vq_enc.", "vq_ref_enc.")))
    assert abs(n - n_vq - 266.355e6) < 1e4
    assert len(folded_param_names()) == len(spec) - sum(k.endswith(".weight_v") for k in spec)


def test_fold_weight_norm_matches_torch():
    torch = pytest.importorskip("torch")
    from detail_tts_amd.weights import fold_weight_norm
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 10, 5))
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 8, 4, padding=2))
    with torch.no_grad():
        conv.weight_g.mul_(1.7)
        convt.weight_g.mul_(0.6)
    sd = {"a." + k: v for k, v in conv.state_dict().items()}
    sd.update({"b." + k: v for k, v in convt.state_dict().items()})
    f = fold_weight_norm(sd)
    x = torch.randn(1, 6, 20)
    with torch.no_grad():
        wa = conv(x)
        ref = torch.nn.functional.conv1d(x, torch.from_numpy(f["a.weight"]), torch.from_numpy(f["a.bias"]))
        assert torch.allclose(wa, ref, atol=1e-5)
        wb = convt(x)
        refb = torch.nn.functional.conv_transpose1d(x, torch.from_numpy(f["b.weight"]), torch.from_numpy(f["b.bias"]), stride=4, padding=2)
        assert torch.allclose(wb, refb, atol=1e-5)


def _packed_conv_ref(x, wp, bp, cout, kw, pad, dil=1, stride=1):
    """numpy emulation of the kernel's GEMM over a packed weight wp[k][CinP][CoutP]."""
    B, cin, T = x.shape
    xp = np.pad(x, ((0, 0), (0, wp.shape[1] - cin), (pad, pad)))
    nout = (T + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    y = np.zeros((B, wp.shape[2], nout), np.float64)
    for tap in range(kw):
        seg = xp[:, :, tap * dil: tap * dil + (nout - 1) * stride + 1: stride]
        y += np.einsum("kc,bkt->bct", wp[tap].astype(np.float64), seg)
    if bp is not None:
        y += bp[None, :, None]
    return y[:, :cout]


def test_pack_conv_layout_against_oracle():
    from detail_tts_amd.packing import pack_conv
    from oracle import ops
    rs = np.random.RandomState(0)
    w, b = rs.randn(37, 21, 5).astype(np.float32), rs.randn(37).astype(np.float32)
    x = rs.randn(2, 21, 33).astype(np.float32)
    wp, bp = pack_conv(w, b)
    assert wp.shape == (5, 32, 64) and bp.shape == (64,)
    np.testing.assert_allclose(_packed_conv_ref(x, wp, bp, 37, 5, 2), ops.conv1d(x, w, b, padding=2), atol=1e-4)


def test_convtranspose_phase_decomposition_against_oracle():
    from detail_tts_amd.packing import convtranspose_as_phases, pack_conv
    from oracle import ops
    rs = np.random.RandomState(1)
    for (cin, cout, k, s, p) in [(10, 6, 16, 8, 4), (7, 5, 8, 4, 2), (9, 4, 2, 2, 0), (5, 3, 6, 2, 2)]:
        w, b = rs.randn(cin, cout, k).astype(np.float32), rs.randn(cout).astype(np.float32)
        x = rs.randn(2, cin, 19).astype(np.float32)
        weq, pad = convtranspose_as_phases(w, s, p)
        wp, bp = pack_conv(weq, np.tile(b, s))
        ref = ops.conv_transpose1d(x, w, b, stride=s, padding=p)
        rows = _packed_conv_ref(x, wp, bp, s * cout, weq.shape[2], pad)        # [B, s*cout, nq]
        nq = rows.shape[2]
        y = rows.reshape(2, s, cout, nq).transpose(0, 2, 3, 1).reshape(2, cout, nq * s)
        assert y.shape == ref.shape
        np.testing.assert_allclose(y, ref, atol=1e-4)


def test_vq_dec_upsample_phases_against_oracle():
    """ConvTranspose1d(k3, s2, p1, output_padding 1) of vq_dec (vqvae/model_24k.py:613-618) = 2 phases x 2 taps, no padding,
    N = T outputs per phase with the input read as zero past its end."""
    from detail_tts_amd.packing import convtranspose_as_phases, pack_conv
    from oracle import vq
    rs = np.random.RandomState(4)
    w, b = rs.randn(9, 5, 3).astype(np.float32), rs.randn(5).astype(np.float32)
    x = rs.randn(2, 9, 11).astype(np.float32)
    weq, pad = convtranspose_as_phases(w, 2, 1, output_padding=1)
    assert weq.shape == (10, 9, 2) and pad == 0
    wp, bp = pack_conv(weq, np.tile(b, 2))
    rows = _packed_conv_ref(np.pad(x, ((0, 0), (0, 0), (0, 1))), wp, bp, 10, 2, 0)          # [B, 2*cout, T]
    y = rows.reshape(2, 2, 5, 11).transpose(0, 2, 3, 1).reshape(2, 5, 22)
    ref = vq.conv_transpose1d_op(x, w, b, 2, 1, 1)
    torch = pytest.importorskip("torch")
    tref = torch.nn.functional.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=2, padding=1,
                                                output_padding=1).numpy()
    np.testing.assert_allclose(ref, tref, atol=1e-5)
    np.testing.assert_allclose(y, ref, atol=1e-4)
    with pytest.raises(ValueError):
        convtranspose_as_phases(w, 2, 1)                                                     # output length != 2T


def test_gate_perm_and_bias_table():
    from detail_tts_amd.packing import bias_table, gate_perm, rel_bucket
    from oracle import diffusion as D
    p = gate_perm(8)
    assert p.tolist() == [0, 4, 1, 5, 2, 6, 3, 7]
    emb = np.random.RandomState(2).randn(32, 16).astype(np.float32)
    tab = bias_table(emb, 48)
    full = D.rel_bias(emb, 200, 48 ** 0.5)            # [H, T, T]
    i, j = np.meshgrid(np.arange(200), np.arange(200), indexing="ij")
    off = np.clip(j - i, -64, 64) + 64
    np.testing.assert_allclose(tab[:, off], full, rtol=0, atol=0)
    assert np.array_equal(rel_bucket(np.arange(-300, 300)), D.rel_bucket(np.arange(-300, 300)))


def test_schedule_mirror_matches_oracle():
    from detail_tts_amd.vqvae.utils.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps
    from oracle import diffusion as D
    d = SpacedDiffusion(space_timesteps(4000, [50]), betas=get_named_beta_schedule("linear", 4000))
    s = D.make_schedule()
    assert d.timestep_map == s["timestep_map"].tolist() and d.num_timesteps == 50
    np.testing.assert_allclose(d.betas, s["betas"], rtol=1e-13)


def test_c_abi_exports_every_declared_symbol():
    """libdetail_hip.so loads without a GPU and exports exactly what include/detail_hip.h declares."""
    from detail_tts_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdetail_hip.so not built (run __graft_entry__.build())")
    hdr = open(os.path.join(ROOT, "include", "detail_hip.h")).read()
    declared = set(re.findall(r"\b(dtts_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dtts_version()
    cfg = _lib.DttsConfig()
    lib.dtts_default_config(ctypes.byref(cfg))
    assert (cfg.diff_channels, cfg.diff_steps, cfg.gpt_mel_codes, cfg.upsample_rates[0]) == (768, 50, 8194, 8)


def test_gpt_options_struct_of_the_ctypes_mirror_matches_the_library():
    """dtts_gpt_options carries its own size: the library refuses a struct of another layout.  dtts_gpt_options_init is
    host-only, so the CPU suite can check that the ctypes mirror (detail_tts_amd/_lib.py) has exactly the layout the shipped library was
    built with - including round 6's token_wgs - and the reference's sampling defaults (vqvae/model_24k.py:782-792)."""
    import ctypes as C
    from detail_tts_amd import _lib
    lib = _lib.load()
    o = _lib.DttsGptOptions()
    lib.dtts_gpt_options_init(C.byref(o))
    assert o.struct_size == C.sizeof(_lib.DttsGptOptions)
    assert (o.max_generate_length, o.top_k, o.token_wgs) == (600, 50, 0)
    assert abs(o.top_p - 0.8) < 1e-6 and abs(o.temperature - 0.8) < 1e-6 and abs(o.repetition_penalty - 2.0) < 1e-6 and o.typical_mass == 0.0
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "detail_hip.h")).read()
    body = hdr[hdr.index("typedef struct dtts_gpt_options {"):hdr.index("} dtts_gpt_options;")]
    for name, _ in _lib.DttsGptOptions._fields_:
        assert name in body, name


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from detail_tts_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.LibraryMissing):
        _lib.load()


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "detail_tts_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)


def test_shard_utterances_balanced():
    from detail_tts_amd.sharding import shard_utterances
    lens = [234] * 64
    sh = shard_utterances(lens, 8)
    assert sorted(sum(sh, [])) == list(range(64)) and all(len(s) == 8 for s in sh)
    sh = shard_utterances([100, 900, 300, 500, 700], 2)
    loads = [sum([100, 900, 300, 500, 700][i] for i in s) for s in sh]
    assert abs(loads[0] - loads[1]) <= 300


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from detail_tts_amd.sharding import broadcast_blob, gather_results, shard_utterances
    blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
    broadcast_blob(blob, src=0)
    mine = shard_utterances([234] * 6, world)[rank]
    allr = gather_results([(i, rank) for i in mine], world)
    q.put((rank, float(blob.sum()), mine, allr))
    dist.destroy_process_group()


def test_two_rank_gloo_weight_broadcast_and_sharding():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert all(abs(r[1] - 499500.0) < 1e-3 for r in res)                 # both ranks hold rank 0's blob
    assert sorted(res[0][2] + res[1][2]) == list(range(6))                 # disjoint cover of the utterances
    assert res[0][3] == res[1][3] and len(sum(res[0][3], [])) == 6         # no data-path collective needed beyond this


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_flag_launches_one_rank_per_gpu(n):
    """SURVEY §8e / BASELINE configs[3]: `python bench.py --gpus N` (2, and the node's 8) with no launcher around it re-execs itself as N
    ranks under torch.distributed.run (127.0.0.1); DTTS_BENCH_LAUNCH_ONLY stops each rank after the rendezvous (gloo here: no GPU in this
    container)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DTTS_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"launched_ranks": n, "n_gpus": n, "local_ranks_seen": n}
    if n != 2:
        return
    # a launcher whose world size contradicts --gpus is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env2, cwd=root)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_voice_bpe_tokenizer_known_answers():
    """Text front-end mirror vs ids produced by the reference tokenizer (tests/golden/tokenizer_kat.json; the first sentence is the
    demo.ipynb KAT: 38 ids).  Needs the reference's vocabulary file, which only exists in the build container."""
    import json
    import os
    vocab = "/root/reference/bpe_tokenizers/zh_tokenizer.json"
    if not os.path.exists(vocab):
        pytest.skip("reference vocabulary file not available on this machine")
    from detail_tts_amd.bpe_tokenizers.voice_tokenizer import VoiceBpeTokenizer, remove_extraneous_punctuation
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tokenizer_kat.json")))
    tok = VoiceBpeTokenizer(vocab)
    assert len(kat[0]["ids"]) == 38
    for case in kat:
        assert tok.encode(case["text"]) == case["ids"]
        assert tok.decode(np.array(case["ids"])) == case["decoded"]
    assert remove_extraneous_punctuation("{a}[b]`c—d") == "(a)(b)'c-d" and remove_extraneous_punctuation("@") == ""


def test_example_api_text_front_end_known_answer():
    """examples/api.py takes the sentence as PINYIN (the output of api.py:21's lazy_pinyin call; pypinyin's dictionary
    cannot be sourced offline) and reproduces api.py:22-24: pad with spaces, VoiceBpeTokenizer.encode -> the demo.ipynb ids (38)."""
    import json
    import subprocess
    import sys
    vocab = "/root/reference/bpe_tokenizers/zh_tokenizer.json"
    if not os.path.exists(vocab):
        pytest.skip("reference vocabulary file not available on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kat = json.load(open(os.path.join(root, "tests", "golden", "tokenizer_kat.json")))[0]
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "api.py"), "--vocab", vocab, "--text", kat["text"].strip(), "--print-ids"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1]) == kat["ids"]


def test_reference_api_imports_resolve_through_compat():
    """compat/README.md: with compat/ first on sys.path every module api.py imports from the reference tree (api.py:10,27,29) - and the
    ones its training / eval code imports - resolves to the MI355X-native mirrors, with the names api.py uses.  (Running api.py itself
    needs a GPU, a checkpoint, torchaudio and pypinyin: none of which this container has.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "from bpe_tokenizers.voice_tokenizer import VoiceBpeTokenizer\n"
        "from prepare.load_infer import load_model\n"
        "from vqvae.utils.data_utils import spectrogram_torch, HParams, mel_spectrogram_torch\n"
        "from vqvae.model_24k import SynthesizerTrn\n"
        "from vqvae.diff_model import DiffusionTts\n"
        "from gpt.model import UnifiedVoice\n"
        "import detail_tts_amd\n"
        "for o in (VoiceBpeTokenizer, load_model, mel_spectrogram_torch, HParams, SynthesizerTrn, DiffusionTts, UnifiedVoice):\n"
        "    assert o.__module__.startswith('detail_tts_amd.'), (o, o.__module__)\n"
        "assert callable(spectrogram_torch) and hasattr(SynthesizerTrn, 'infer') and hasattr(UnifiedVoice, 'inference_speech_tortoise')\n"
        "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(root, "compat"), root]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_configs0_cpu_plumbing_on_bundled_prompt(weights):
    """BASELINE configs[0]: the api.py flow on the reference's bundled 1.wav (44.1 kHz mono int16, 195 979 samples) with the
    demo.ipynb pinyin sentence (38 ids + trailing 0), random-init weights, on the CPU oracle: resample -> 416 mel frames -> 3 codes
    -> 12 mel frames -> 3072 samples.  Plumbing check of the whole path at the prompt length api.py really uses."""
    import json
    import wave
    from oracle import frontend as FE, pipeline
    here = os.path.dirname(__file__)
    with wave.open(os.path.join(here, "golden", "prompt_1.wav"), "rb") as f:
        sr, n = f.getframerate(), f.getnframes()
        pcm = np.frombuffer(f.readframes(n), np.int16).reshape(-1, f.getnchannels())
    assert (sr, n) == (44100, 195979)
    audio = FE.resample(pcm[:, 0].astype(np.float32)[None] / 32768.0, sr, 24000)
    assert audio.shape[1] == 106656
    mel = FE.mel_spectrogram(audio)[0]
    assert mel.shape == (128, 416)
    ids = json.load(open(os.path.join(here, "golden", "tokenizer_kat.json")))[0]["ids"]
    text = np.array(ids + [0], np.int64)                       # F.pad(text_tokens, (0, 1)), api.py:25
    assert len(ids) == 38
    wav = pipeline.infer_one(weights, text, mel, 1234, 0, max_generate_length=4, suppress_eos=True, diffusion_steps=2)
    assert wav.shape == (3 * 1024,) and np.isfinite(wav).all() and float(np.abs(wav).max()) > 1e-4


def test_shipped_library_has_no_packed_fp32_math():
    """The WHOLE library is built without packed fp32 VALU instructions (build.py: NO_PACKED_FP32; csrc/gpt_token.hip additionally with
    -fno-slp-vectorize): with v_pk_{fma,mul,add}_f32 the GPT token kernel's results were wrong when its waves shared SIMDs with the
    split-precision kernels' fp16 MFMAs (DESIGN.md par. 4, profiles/r04_token_pk_diag.txt), and every kernel of stages A / B / C runs next
    to those kernels under SynthesizerTrn.infer_stream.  The SHIPPED .so is disassembled - the file the GPU box loads - not build/*.o."""
    import subprocess
    import tempfile
    from detail_tts_amd import build as B
    assert "-fno-slp-vectorize" in B._extra_flags("gpt_token.hip")
    assert "-packed-fp32-ops" in B.FLAGS
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    assert os.path.exists(objdump), "llvm-objdump is part of the ROCm image"
    assert os.path.exists(B.LIB), "libdetail_hip.so must be built in-tree (python -m detail_tts_amd.build)"
    import shutil
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(B.LIB, so)
        subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=td)
        dev = [f for f in os.listdir(td) if "gfx950" in f]
        assert dev, os.listdir(td)
        asm = "".join(subprocess.run([objdump, "-d", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout for f in dev)
    for kernel in ("gpt_token_kernel", "conv_x3_kernel", "flash_attn_x3", "conv_x3d_kernel", "resblock1x3_fused", "gemv_block_kernel",
                   "gn_split_planes", "conv_gemm_kernel"):
        assert kernel in asm, kernel
    assert "v_fma_f32" in asm and "v_mfma_f32_32x32x16_f16" in asm
    packed = [l for l in asm.splitlines() if "v_pk_fma_f32" in l or "v_pk_mul_f32" in l or "v_pk_add_f32" in l]
    assert not packed, (len(packed), packed[:5])
