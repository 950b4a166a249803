"""Timeline of SynthesizerTrn.infer_stream at the bench workload (configs[2]: batch 8, 10 s prompts, 234 codes): per request, when each
stage was ENQUEUED (host clock) and when it RAN (stream events), all relative to the first request's start.

    python tools/pipeline_trace.py [--requests 6] [--batch 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--codes", type=int, default=234)
    ap.add_argument("--reserve", type=int, default=0, help="mask the diffusion streams off the last N CUs (left to stage A)")
    args = ap.parse_args()
    main_stream = None
    if args.reserve:
        import ctypes
        os.environ["DTTS_B_CU_RESERVE"] = str(args.reserve)
        hip = ctypes.CDLL("libamdhip64.so")
        mask = (ctypes.c_uint32 * 8)(*([0xffffffff] * 8))
        for b in range(256 - args.reserve, 256):
            mask[b >> 5] &= ~(1 << (b & 31))
        st = ctypes.c_void_p()
        torch.cuda.init()
        assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, mask) == 0
        main_stream = torch.cuda.ExternalStream(st.value)
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
    model = SynthesizerTrn(select_inference_params(synthetic_state_dict(0)), folded=True)
    B, n = args.batch, args.codes
    rs = np.random.RandomState(1)
    refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
    text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 60)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
    tl, rl = torch.full((B,), 61), torch.full((B,), 936)

    def reqs(k, first):
        return (dict(text=text, text_length=tl, refer=refer, refer_lengths=rl, seed=first + i, sample_ids=list(range(B))) for i in range(k))

    torch.cuda.synchronize()
    ctx = torch.cuda.stream(main_stream) if main_stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        list(model.infer_stream(reqs(2, 0), max_generate_length=n + 1, suppress_eos=True))       # warm-up
        torch.cuda.synchronize()
        model.stream_trace = []
        t0 = time.perf_counter()
        out = list(model.infer_stream(reqs(args.requests, 10), max_generate_length=n + 1, suppress_eos=True))
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr = model.stream_trace
    model.stream_trace = None
    e0, h0 = tr[0]["ev_a0"], tr[0]["host_a0"]
    print(f"{args.requests} requests of {B} x {n} codes: {dt * 1e3 / args.requests:.1f} ms per request, "
          f"{args.requests * B * n * 1024 / 24000.0 / dt:.1f} audio-s/s")
    print("req |   host: A enqueue      A results   B+C enqueue     |   device: stage A          stage B          stage C end")
    for i, t in enumerate(tr):
        h = lambda k: (t[k] - h0) * 1e3
        d = lambda k: e0.elapsed_time(t[k])
        print(f"{i:3d} | {h('host_a0'):7.1f}..{h('host_a1'):7.1f}   {h('host_a2'):7.1f}   {h('host_b0'):7.1f}..{h('host_b1'):7.1f} | "
              f"{d('ev_a0'):7.1f}..{d('ev_a1'):7.1f} ({d('ev_a1') - d('ev_a0'):6.1f})  {d('ev_b0'):7.1f}..{d('ev_b1'):7.1f} ({d('ev_b1') - d('ev_b0'):6.1f})  "
              f"{d('ev_c1'):7.1f}")
    assert all(bool(torch.isfinite(w).all()) for w, _ in out)


if __name__ == "__main__":
    main()
