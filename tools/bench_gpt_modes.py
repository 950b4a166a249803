"""Host cost of the GPT decode launch modes: wall time of each dtts_gpt_decode call (graph replay) / decode_step (eager) vs the
GPU time of the steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm
rt = Runtime(fold_weight_norm(synthetic_state_dict(0, only_prefixes=["gpt."])), folded=True, parts=("gpt",))
rs = np.random.RandomState(1)
B, G = 8, 235
refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
texts = [np.concatenate([rs.randint(3, 255, 60), [0]]) for _ in range(B)]
use_stream = os.environ.get("SIDE_STREAM") == "1"
st = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
with torch.cuda.stream(st):
    for mode in ("graph16", "graph16", "eager", "graph1"):
        rt.gpt_prefill(refer, None, texts, 1, list(range(B)), max_generate_length=G, suppress_eos=True)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        if mode == "graph16":
            while rt.gpt_steps() < G:
                t = time.perf_counter(); rt.gpt_decode(16); host.append(time.perf_counter() - t)
        elif mode == "graph1":
            while rt.gpt_steps() < G:
                t = time.perf_counter(); rt.gpt_decode(1); host.append(time.perf_counter() - t)
        else:
            while rt.gpt_steps() < G:
                t = time.perf_counter(); rt.gpt_decode_step(); host.append(time.perf_counter() - t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rt.gpt_finish()
        print(f"{mode:8s} calls {len(host):4d}  host per call {np.mean(host)*1e6:9.1f} us (first {host[0]*1e6:9.1f})  enqueue {1e3*(t1-t0):7.2f} ms  total {1e3*(t2-t0):7.2f} ms")
