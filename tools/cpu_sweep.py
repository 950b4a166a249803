import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
from oracle import diffusion as D, ops
W = select_inference_params(synthetic_state_dict(0))
ops.use_torch(True)
rs = np.random.RandomState(1)
code_emb = rs.randn(1, 768, 936).astype(np.float32); x = rs.randn(1, 128, 936).astype(np.float32)
sched = D.make_schedule()
D.diffusion_forward(W, x, [sched["timestep_map"][49]], code_emb)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=nt)
    except Exception:
        ctx = None
    t0 = time.time()
    for _ in range(2):
        D.diffusion_forward(W, x, [sched["timestep_map"][25]], code_emb)
    print(nt, "threads:", (time.time() - t0) / 2, "s per forward", flush=True)
