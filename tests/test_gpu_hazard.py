"""Co-residency gates for the WHOLE library (VERDICT r04 "weak" 1 / "next" 2).

Round 3 found that the persistent GPT token kernel, built with packed fp32 VALU math, returned wrong accumulators whenever its waves
shared a SIMD with waves issuing dense fp16 MFMAs (profiles/r04_token_pk_diag.txt); round 4 removed packed math from that ONE object.
Every other kernel runs next to the split-precision trunk too: stage C and stage A of other requests under stage B in
SynthesizerTrn.infer_stream, the two CFG chunks of stage B next to each other.  These tests hold each stage bit-identical with and
without the other stages running on the chip, at the headline size, and the library itself is now built without packed fp32
instructions (tests/test_host_logic.py disassembles the shipped .so).

What every request must equal: vqvae/model_24k.py:774-810 (one blocking infer)."""
import threading
import time

import numpy as np
import pytest

from fullsize_inputs import N_CODES, T

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

T_REF, L_TEXT = T, 60


@pytest.fixture(scope="module")
def synth():
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
    return SynthesizerTrn(select_inference_params(synthetic_state_dict(0, variant="signal")), folded=True)


def _requests(n, B=8):
    rs = np.random.RandomState(17)
    out = []
    for i in range(n):
        refer = torch.from_numpy((rs.randn(B, 128, T_REF) * 2 - 5).astype(np.float32)).cuda()
        text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, L_TEXT)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
        out.append(dict(text=text, text_length=torch.full((B,), L_TEXT + 1), refer=refer, refer_lengths=torch.full((B,), T_REF),
                        seed=9000 + i, sample_ids=[100 * i + b for b in range(B)]))
    return out


def test_headline_size_infer_stream_equals_blocking_infer(synth):
    """Eight headline requests (8 utterances x 234 sampled codes x 50 sampling steps, signal weights: the waveform depends on the whole
    path) through the three-stream pipeline - three requests' stages overlap on the chip for seconds - against one blocking infer() per
    request: same lengths, same samples, bit for bit."""
    reqs = _requests(8)
    G = N_CODES + 1
    outs = list(synth.infer_stream(iter(reqs), max_generate_length=G, suppress_eos=True))
    assert len(outs) == len(reqs)
    for i, (r, (wav, lens)) in enumerate(zip(reqs, outs)):
        ref, rlens = synth.infer(r["text"], r["text_length"], r["refer"], r["refer_lengths"], batch=True, seed=r["seed"],
                                 sample_ids=r["sample_ids"], max_generate_length=G, suppress_eos=True, return_lengths=True)
        assert lens == rlens == [N_CODES * 1024] * 8
        assert float(ref.pow(2).mean().sqrt()) > 0.05                     # the signal weights do what they are for
        assert torch.equal(wav, ref), f"request {i}: pipelined waveform differs from the blocking one (max-abs {float((wav - ref).abs().max()):.3e})"


# ---- one stage repeated under another stage's load ------------------------------------------------------------------------------------

def _trunk_load(rt):
    """stage B at the headline shape: both CFG chunks of conv_x3 / flash_attn_x3w / gn_split_planes at B = 8, T = 936"""
    r8 = torch.from_numpy((np.random.RandomState(1).randn(8, 128, T_REF) * 2 - 5).astype(np.float32)).cuda()
    ce = rt.diff_timestep_independent(torch.randn(8, 768, N_CODES, device="cuda"), rt.diff_conditioning(r8))
    return lambda: rt.diff_sample(ce, 3, list(range(8)), n_steps=3)


def _vocoder_load(rt):
    mel = torch.from_numpy((np.random.RandomState(2).randn(8, 128, T) * 2 - 5).astype(np.float32)).cuda()
    return lambda: rt.vocoder(mel, 3, list(range(8)))


def _decode_load(rt):
    rs = np.random.RandomState(3)
    refer = torch.from_numpy((rs.randn(8, 128, 300) * 2 - 5).astype(np.float32)).cuda()
    texts = [np.concatenate([rs.randint(3, 255, 20), [0]]).astype(np.int32) for _ in range(8)]
    return lambda: rt.gpt_generate(refer, None, texts, 5, list(range(8)), max_generate_length=48, suppress_eos=True)


def _target_vocoder(rt):
    """stage C: enc_p / flow on fp32 MFMA, conv_x3d, the LDS-resident fused ResBlock1, polyphase upsamplers"""
    mel = torch.from_numpy((np.random.RandomState(4).randn(8, 128, 400) * 2 - 5).astype(np.float32)).cuda()
    return lambda: (rt.vocoder(mel, 7, list(range(8))).clone(),)


def _target_chain_decode(rt):
    """stage A on the launch-per-GEMV chain (17+-row sessions, graph mode, and the token kernel's time-out replay)"""
    rs = np.random.RandomState(5)
    refer = torch.from_numpy((rs.randn(3, 128, 200) * 2 - 5).astype(np.float32)).cuda()
    texts = [np.concatenate([rs.randint(3, 255, 10), [0]]).astype(np.int32) for _ in range(3)]

    def gen():
        c, n, l = rt.gpt_generate(refer, None, texts, 5, [0, 1, 2], max_generate_length=24, suppress_eos=True)
        return torch.from_numpy(np.ascontiguousarray(c)), l.clone()
    return gen


def _target_trunk(rt):
    """stage B itself (three sampling steps at B = 8, T = 400) - under stage C / stage A of other requests"""
    r8 = torch.from_numpy((np.random.RandomState(6).randn(8, 128, 300) * 2 - 5).astype(np.float32)).cuda()
    ce = rt.diff_timestep_independent(torch.randn(8, 768, 100, device="cuda", generator=torch.Generator("cuda").manual_seed(1)),
                                      rt.diff_conditioning(r8))
    return lambda: (rt.diff_sample(ce, 3, list(range(8)), n_steps=3).clone(),)


# (target, load, repetitions, options, how the load is issued).  The library's thread contract (include/detail_hip.h "Threads") allows two
# host threads per handle only as {decode-session entry points | everything else}: a decode target / load runs against the other
# stages from a second thread (infer_stream's split); stage B against stage C is issued from ONE thread on two streams, as
# infer_stream does.
CASES = {
    "vocoder_under_trunk": (_target_vocoder, _trunk_load, 200, {}, "stream"),
    "chain_decode_under_trunk": (_target_chain_decode, _trunk_load, 200, {"gpt_token_kernel": 0}, "thread"),
    "trunk_under_vocoder": (_target_trunk, _vocoder_load, 100, {}, "stream"),
    "trunk_under_token_decode": (_target_trunk, _decode_load, 100, {}, "thread"),
}


@pytest.mark.parametrize("case", list(CASES))
def test_stage_repeated_under_another_stages_load_is_bit_identical(synth, case):
    """`target` run alone, then N times while `load` keeps running on another stream: every repetition must equal the lone run bit
    for bit (torch.equal)."""
    rt = synth.rt
    target_f, load_f, n, opts, how = CASES[case]
    for k, v in opts.items():
        rt.set_option(k, v)
    try:
        target = target_f(rt)
        ref = target()
        torch.cuda.synchronize()
        bad = 0
        if how == "stream":
            s_load = torch.cuda.Stream()
            with torch.cuda.stream(s_load):
                body = load_f(rt)
                body()
            for _ in range(n):
                with torch.cuda.stream(s_load):          # enqueued, not waited for: the target's kernels join it on the chip
                    body()
                out = target()
                bad += not all(torch.equal(a, b) for a, b in zip(ref, out))
            torch.cuda.synchronize()
        else:
            stop = threading.Event()
            failed, rounds = [], [0]

            def run_load():
                try:
                    torch.cuda.set_device(0)
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        body = load_f(rt)
                        while not stop.is_set():
                            body()
                            s.synchronize()
                            rounds[0] += 1
                except Exception as e:      # pragma: no cover
                    failed.append(e)

            th = threading.Thread(target=run_load)
            th.start()
            try:
                time.sleep(1.0)
                for _ in range(n):
                    out = target()
                    bad += not all(torch.equal(a, b) for a, b in zip(ref, out))
            finally:
                stop.set()
                th.join()
            assert not failed, failed
            assert rounds[0] >= 2, "the load did not run next to the target"
        assert bad == 0, f"{case}: {bad} of {n} repetitions differ from the run alone"
    finally:
        for k in opts:
            rt.set_option(k, 1)
