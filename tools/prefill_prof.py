"""Per-kernel timings of stage A's prefill (conditioning encoder + GPT prefill + the first token), library profiler.  BB = rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm
rt = Runtime(fold_weight_norm(synthetic_state_dict(0, only_prefixes=["gpt."])), folded=True, parts=("gpt",))
rs = np.random.RandomState(1)
B = int(os.environ.get("BB", 8))
refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
texts = [np.concatenate([rs.randint(3, 255, 60), [0]]) for _ in range(B)]
for _ in range(3):
    rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=1, suppress_eos=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=1, suppress_eos=True)
torch.cuda.synchronize(); print(f"B={B} G=1: {(time.perf_counter()-t)/5*1e3:.2f} ms")
rt.profile_enable(2)
rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=1, suppress_eos=True)
torch.cuda.synchronize()
tot = 0
for p in sorted(rt.profile_report(), key=lambda p: -p["total_ms"])[:16]:
    tot += p["total_ms"]
    print("%-44s %5d launches %8.3f ms  %7.1f us" % (p["name"], p["launches"], p["total_ms"], p["total_ms"] / p["launches"] * 1e3))
print("sum of listed: %.2f ms" % tot)
