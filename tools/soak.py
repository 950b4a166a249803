"""Soak of the three-stream request pipeline at the board's power limit.

The packed-fp32 corruption of rounds 3 / 4 (profiles/r04_token_pk_diag.txt) was closed by REMOVING the instructions, not by explaining it,
and round 5 showed that the chip sits at its power / current limit exactly where it appeared.  The co-residency gates of
tests/test_gpu_hazard.py are minutes of exposure.  This tool is the long run: REQUESTS headline requests (8 utterances x 234 sampled
codes x 50 sampling steps, the signal weights: the waveform depends on the whole path) through SynthesizerTrn.infer_stream, EVERY
pipelined waveform compared bit for bit (on the device, on a side stream, so that the pipeline never drains) with the blocking infer()
of the same request - the DISTINCT distinct requests are made once, blocking, before the run and cycle.  Once per setting of
gpt_token_exclusive_cu (1: the token kernel asks for its CU's whole LDS; 0: it shares the CU with the trunk's workgroups).

What every request must equal: vqvae/model_24k.py:774-810.

    python tools/soak.py > profiles/r06_soak.txt        (REQUESTS = 500, DISTINCT = 10, SETTINGS = "1,0")
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PowerSampler   # noqa: E402
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn   # noqa: E402
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict   # noqa: E402

REQUESTS = int(os.environ.get("REQUESTS", 500))
DISTINCT = int(os.environ.get("DISTINCT", 10))
SETTINGS = [int(v) for v in os.environ.get("SETTINGS", "1,0").split(",")]
B, N_CODES, T_REF, L_TEXT = 8, 234, 936, 60
G = N_CODES + 1


def requests(n):
    rs = np.random.RandomState(17)
    out = []
    for i in range(n):
        refer = torch.from_numpy((rs.randn(B, 128, T_REF) * 2 - 5).astype(np.float32)).cuda()
        text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, L_TEXT)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
        out.append(dict(text=text, text_length=torch.full((B,), L_TEXT + 1), refer=refer, refer_lengths=torch.full((B,), T_REF),
                        seed=9000 + i, sample_ids=[100 * i + b for b in range(B)]))
    return out


def main():
    m = SynthesizerTrn(select_inference_params(synthetic_state_dict(0, variant="signal")), folded=True)
    reqs = requests(DISTINCT)
    refs = []
    for r in reqs:
        w = m.infer(r["text"], r["text_length"], r["refer"], r["refer_lengths"], batch=True, seed=r["seed"], sample_ids=r["sample_ids"],
                    max_generate_length=G, suppress_eos=True)
        assert float(w.pow(2).mean().sqrt()) > 0.05
        refs.append(w)
    again = m.infer(reqs[0]["text"], reqs[0]["text_length"], reqs[0]["refer"], reqs[0]["refer_lengths"], batch=True, seed=reqs[0]["seed"],
                    sample_ids=reqs[0]["sample_ids"], max_generate_length=G, suppress_eos=True)
    assert torch.equal(again, refs[0]), "the blocking call itself is not reproducible"
    print(f"soak: {REQUESTS} pipelined headline requests per setting ({B} x {N_CODES} codes, 50 sampling steps, signal weights), "
          f"{DISTINCT} distinct requests cycling, every waveform torch.equal to its blocking infer(); settings gpt_token_exclusive_cu = {SETTINGS}", flush=True)
    side = torch.cuda.Stream()
    total_bad = 0
    for setting in SETTINGS:
        m.rt.set_option("gpt_token_exclusive_cu", setting)
        flags = []
        power = PowerSampler(0).start()
        t0 = time.perf_counter()
        tl = t0
        gen = m.infer_stream((reqs[i % DISTINCT] for i in range(REQUESTS)), max_generate_length=G, suppress_eos=True)
        for i, (wav, lens) in enumerate(gen):
            assert lens == [N_CODES * 1024] * B
            with torch.cuda.stream(side):                     # the waveform is complete (infer_stream waited for it): compare beside the pipeline
                flags.append((wav == refs[i % DISTINCT]).all())
                wav.record_stream(side)
            if (i + 1) % 50 == 0:
                side.synchronize()
                bad = [j for j, f in enumerate(flags) if not bool(f)]
                now = time.perf_counter()
                print(f"  exclusive_cu={setting}: {i + 1:4d} requests, {(now - tl) / 50 * 1e3:7.1f} ms per request, mismatches so far {len(bad)} {bad[:8]}", flush=True)
                tl = now
        side.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pw = power.stop()
        bad = [j for j, f in enumerate(flags) if not bool(f)]
        total_bad += len(bad)
        print(f"exclusive_cu={setting}: {REQUESTS} requests in {dt:.1f} s ({dt / REQUESTS * 1e3:.1f} ms per request), {len(bad)} mismatching waveforms {bad[:16]}; "
              f"saturated requests re-run on fp32: {m.saturated_requests}; board power {pw}", flush=True)
    m.rt.set_option("gpt_token_exclusive_cu", 0)
    print("SOAK CLEAN" if total_bad == 0 else f"SOAK FOUND {total_bad} MISMATCHES")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
