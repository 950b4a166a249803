"""Per-kernel timings of stage C (vocoder) at the bench workload: 8 utterances of 936 mel frames.   BB / TT override."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict

B, T = int(os.environ.get("BB", 8)), int(os.environ.get("TT", 936))
rt = Runtime(select_inference_params(synthetic_state_dict(0)), folded=True, parts=("vocoder",))
mel = torch.from_numpy((np.random.RandomState(5).randn(B, 128, T) * 2 - 5).astype(np.float32)).cuda()
ids = list(range(B))
rt.vocoder(mel, 1, ids)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    rt.vocoder(mel, 1, ids)
e1.record()
torch.cuda.synchronize()
print(f"stage C, B = {B}, T = {T}: {e0.elapsed_time(e1) / 5:.2f} ms per call")
rt.profile_enable(2)
rt.vocoder(mel, 1, ids)
torch.cuda.synchronize()
for p in sorted(rt.profile_report(), key=lambda p: -p["total_ms"])[:14]:
    print("%-40s %4d launches %8.2f ms  %8.1f us  %7.1f TFLOP/s(eq) %6.2f TB/s(alg)" % (
        p["name"], p["launches"], p["total_ms"], p["total_ms"] / p["launches"] * 1e3, p["flops"] / max(p["total_ms"], 1e-9) / 1e9,
        p["bytes"] / max(p["total_ms"], 1e-9) / 1e9))
