"""Stage-A timing: prefill + 234 decode steps at batch 8 (10 s prompt, 60 text ids)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
W = select_inference_params(synthetic_state_dict(0, only_prefixes=["gpt."])) if False else None
from detail_tts_amd.weights import inference_param_spec
sd = synthetic_state_dict(0, only_prefixes=["gpt."])
from detail_tts_amd.weights import fold_weight_norm
rt = Runtime(fold_weight_norm(sd), folded=True, parts=("gpt",))
rs = np.random.RandomState(1)
B = int(os.environ.get("BB", 8))
refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
texts = [np.concatenate([rs.randint(3, 255, 60), [0]]) for _ in range(B)]
for G in (2, 235):
    rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=G, suppress_eos=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2): rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=G, suppress_eos=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
    print(f"B={B} G={G}: {dt*1e3:.1f} ms")
if os.environ.get("PROF") == "1":
    rt.profile_enable(2)
    rt.gpt_generate(refer, None, texts, 1, list(range(B)), max_generate_length=65, suppress_eos=True)
    torch.cuda.synchronize()
    for p in sorted(rt.profile_report(), key=lambda p: -p["total_ms"])[:10]:
        print("%-44s %5d launches %8.2f ms  %7.1f us" % (p["name"], p["launches"], p["total_ms"], p["total_ms"] / p["launches"] * 1e3))
