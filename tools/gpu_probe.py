"""Ad-hoc GPU probe: parity numbers + first timings (not part of the test suite)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
t0 = time.time()
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
from detail_tts_amd.runtime import Runtime
from oracle import diffusion as D
W = select_inference_params(synthetic_state_dict(0)); t1 = time.time()
rt = Runtime(W, folded=True, parts=("diffusion",)); torch.cuda.synchronize(); t2 = time.time()
print(f"cpu cores {os.cpu_count()} synth {t1-t0:.1f}s pack+bind {t2-t1:.1f}s", torch.cuda.get_device_name(0))
g = dict(np.load("tests/golden/diff_forward.npz"))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
oc = rt.diff_forward(dev(g["x"]), 47, dev(g["code_emb"])).cpu().numpy()
print("diff_forward vs golden maxabs", np.abs(oc - g["out_cond"]).max(), "ref scale", np.abs(g["out_cond"]).max())
# timing at the headline size
for B, T in ((1, 936), (8, 936)):
    x = torch.randn(B, 128, T, device="cuda"); ce = torch.randn(B, 768, T, device="cuda")
    for _ in range(2): rt.diff_forward(x, 30, ce)
    torch.cuda.synchronize(); t = time.time(); n = 5
    for _ in range(n): rt.diff_forward(x, 30, ce)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    fl = 2 * B * 166.97e9
    print(f"B={B} T={T}: forward pair {dt*1e3:.2f} ms -> {fl/dt/1e12:.1f} TFLOP/s (algorithmic)")
