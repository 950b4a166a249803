cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
REPS=2 bash tools/batch1_ab.sh "DTTS_ATTN_KSPLIT=1" "DTTS_ATTN_KSPLIT=2" "DTTS_ATTN_KSPLIT=3" "DTTS_ATTN_KSPLIT=4" 2>&1 | tee gpurun_out/b1_ksplit.txt
# per-kernel profile of one blocking batch-1 request
for ks in 1 2; do
DTTS_ATTN_KSPLIT=$ks python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b1_prof_ks$ks.txt
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
m = SynthesizerTrn(select_inference_params(synthetic_state_dict(0)), folded=True)
rs = np.random.RandomState(1)
B, n = 1, 234
req = dict(text=torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 60)), np.zeros((B, 1), np.int64)], 1)), text_length=torch.full((B,), 61),
           refer=torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda(), refer_lengths=torch.full((B,), 936))
kw = dict(max_generate_length=n + 1, suppress_eos=True, batch=True, seed=1)
for _ in range(2): m.infer(**req, **kw)
torch.cuda.synchronize()
m.rt.profile_enable(2)
m.infer(**req, **kw)
torch.cuda.synchronize()
rep = sorted(m.rt.profile_report(), key=lambda p: -p["total_ms"])
tot = sum(p["total_ms"] for p in rep)
print(f"ksplit={os.environ.get('DTTS_ATTN_KSPLIT')}: {tot:.1f} ms of bracketed kernels in one batch-1 request")
for p in rep[:22]:
    print("%-46s %6d launches %8.2f ms  %7.1f us" % (p["name"], p["launches"], p["total_ms"], p["total_ms"] / p["launches"] * 1e3))
PY
done
