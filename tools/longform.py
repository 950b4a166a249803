"""Long-form generation (BASELINE configs[4]): 60 s utterances (n = 1406 codes, T = 5624 mel frames), batch 4 per GPU, the vocoder's
generator streamed window by window on its own HIP stream.  Prints one `infer` call with per-stage times and peak device memory, then a
pipelined run of several such requests (`SynthesizerTrn.infer_stream`: stage A of request i+1 and stage C of request i under the
diffusion of request i / i+1).

    python tools/longform.py            (BB = batch, NN = codes, REQ = pipelined requests, CHUNK = generator window in mel frames)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict

B, N = int(os.environ.get("BB", 4)), int(os.environ.get("NN", 1406))
REQ, CHUNK = int(os.environ.get("REQ", 4)), int(os.environ.get("CHUNK", 256))
m = SynthesizerTrn(select_inference_params(synthetic_state_dict(0)), folded=True)
rs = np.random.RandomState(1)
refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 60)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
tl, rl = torch.full((B,), 61), torch.full((B,), 936)
audio = B * N * 1024 / 24000.0

kw = dict(batch=True, sample_ids=list(range(B)), max_generate_length=N + 1, suppress_eos=True, return_lengths=True)
m.infer(text, tl, refer, rl, seed=0, stream_vocoder=True, vocoder_chunk=CHUNK, **kw)           # warm-up (arena growth)
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
free0, total = torch.cuda.mem_get_info()
m.stage_ms = {}
t = time.perf_counter()
wav, lens = m.infer(text, tl, refer, rl, seed=1, stream_vocoder=True, vocoder_chunk=CHUNK, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
stages = {k: round(v) for k, v in m.stage_ms.items()}
m.stage_ms = None
free1, _ = torch.cuda.mem_get_info()
print(f"one infer() call, B = {B}, n = {N} codes (T = {4 * N} frames, {N * 1024 / 24000:.1f} s each), generator window {CHUNK} frames:")
print(f"  {dt:.2f} s for {audio:.1f} s of audio -> {audio / dt:.1f} audio-s/s; finite = {bool(torch.isfinite(wav).all())}, "
      f"rms = {float(wav.pow(2).mean().sqrt()):.4f}; stage ms = {stages}")
print(f"  device memory in use: {(total - min(free0, free1)) / 2**30:.1f} GiB of {total / 2**30:.0f} (library arenas + torch), "
      f"torch peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")

reqs = (dict(text=text, text_length=tl, refer=refer, refer_lengths=rl, seed=10 + i, sample_ids=list(range(B))) for i in range(REQ))
torch.cuda.synchronize()
t = time.perf_counter()
outs = list(m.infer_stream(reqs, max_generate_length=N + 1, suppress_eos=True, vocoder_chunk=CHUNK))
torch.cuda.synchronize()
dt = time.perf_counter() - t
assert all(bool(torch.isfinite(w).all()) for w, _ in outs)
print(f"infer_stream, {REQ} such requests: {dt / REQ:.2f} s per request -> {REQ * audio / dt:.1f} audio-s/s")
