import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from detail_tts_amd.runtime import Runtime
from detail_tts_amd.weights import synthetic_state_dict, fold_weight_norm
rt = Runtime(fold_weight_norm(synthetic_state_dict(0, only_prefixes=["gpt."])), folded=True, parts=("gpt",))
rs = np.random.RandomState(3)
B = 8
r1 = (rs.randn(1, 128, 120) * 2 - 5).astype(np.float32)
refer = torch.from_numpy(np.repeat(r1, B, 0)).cuda()
t1 = np.concatenate([rs.randint(3, 255, 7), [0]]).astype(np.int32)
texts = [t1.copy() for _ in range(B)]
forced = [np.array([100, 200, 300], np.int32) for _ in range(B)]
c, n, l = rt.gpt_generate(refer, None, texts, 77, [5] * B, max_generate_length=3, suppress_eos=True, forced_codes=forced)
print("latent row diffs vs row 0:", [float((l[b] - l[0]).abs().max()) for b in range(B)])
