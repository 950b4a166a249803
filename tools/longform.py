"""Long-form sanity: 60 s generation (n = 1406 codes, T = 5624 frames), batch B (configs[4] of BASELINE.json)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
B, N = int(os.environ.get("BB", 1)), int(os.environ.get("NN", 1406))
m = SynthesizerTrn(select_inference_params(synthetic_state_dict(0)), folded=True)
rs = np.random.RandomState(1)
refer = torch.from_numpy((rs.randn(B, 128, 936) * 2 - 5).astype(np.float32)).cuda()
text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, 60)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
m.stage_ms = {}
t = time.perf_counter()
wav, lens = m.infer(text, torch.full((B,), 61), refer, torch.full((B,), 936), batch=True, seed=1, sample_ids=list(range(B)),
                    max_generate_length=N + 1, suppress_eos=True, return_lengths=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"B={B} n={N}: {dt:.2f}s for {B*N*1024/24000:.1f}s audio -> {B*N*1024/24000/dt:.1f}x realtime; finite={bool(torch.isfinite(wav).all())} "
      f"rms={float(wav.pow(2).mean().sqrt()):.4f} stages={ {k: round(v) for k, v in m.stage_ms.items()} }")
