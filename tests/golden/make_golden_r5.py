#!/usr/bin/env python3
"""Round-5 fixtures FROM THE REFERENCE ITSELF (build container only, CPU): the off-path branches of
UnifiedVoice.inference_speech_tortoise (gpt/model.py:514-545) that SynthesizerTrn.infer never takes and the mirror used to refuse -
greedy decoding (do_sample=False), num_return_sequences > 1 (HF expands the batch by repeat_interleave), input_tokens (mel tokens
in front of the generated ones) and typical sampling (TypicalLogitsWarper: HF runs it between the repetition penalty and the temperature).  Same small inputs as make_golden.py's `gpt_generate`, the sampler's multinomial patched to the Philox
spec (row b of the expanded batch draws from stream sample_id + b).  Stores inputs and the reference's codes only.

    python tests/golden/make_golden_r5.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import SEED_N, build_reference_model, install_shim, philox_rng, save   # noqa: E402


def main():
    install_shim()
    import torch
    torch.set_grad_enabled(False)
    m = build_reference_model()
    g = m.gpt
    rs = np.random.RandomState(1)
    T_ref, L0 = 64, 12
    refer = (rs.randn(1, 128, T_ref) * 2 - 5).astype(np.float32)
    text = np.concatenate([rs.randint(3, 255, (1, L0)), [[0]]], 1).astype(np.int32)
    refer_t, text_t, rl = torch.from_numpy(refer), torch.from_numpy(text), torch.tensor([T_ref])
    kw = dict(top_p=0.8, temperature=0.8, length_penalty=1.0, repetition_penalty=2.0, max_generate_length=10)
    out = {}
    # greedy: only the repetition penalty is applied (the warpers belong to sampling), argmax
    out["greedy"] = g.inference_speech_tortoise(refer_t, rl, text_t, do_sample=False, num_return_sequences=1, length_penalty=1.0,
                                                repetition_penalty=2.0, max_generate_length=10).numpy()
    with philox_rng(sample_id=7):
        out["nrs2"] = g.inference_speech_tortoise(refer_t, rl, text_t, do_sample=True, num_return_sequences=2, **kw).numpy()
    input_tokens = np.array([[5, 77, 4001]], np.int64)
    with philox_rng(sample_id=7) as st:
        st["gpt_step"] = input_tokens.shape[1]          # the noise spec keys a draw by its mel position (oracle/philox.py): forced positions draw nothing
        out["input_tokens_codes"] = g.inference_speech_tortoise(refer_t, rl, text_t, input_tokens=torch.from_numpy(input_tokens), do_sample=True,
                                                                num_return_sequences=1, **kw).numpy()
    # input_tokens AND num_return_sequences = n (single prompt only: the reference's torch.cat needs n * B == n): the reference tiles
    # the prefixes to n rows and HF expands every row n times again -> n * n rows, row r starts with input_tokens[(r // n) % rows]
    input_tokens2 = np.array([[5, 77, 4001], [900, 13, 2]], np.int64)
    with philox_rng(sample_id=7) as st:
        st["gpt_step"] = input_tokens2.shape[1]
        out["input_tokens_nrs2_codes"] = g.inference_speech_tortoise(refer_t, rl, text_t, input_tokens=torch.from_numpy(input_tokens2), do_sample=True,
                                                                     num_return_sequences=2, **kw).numpy()
    with philox_rng(sample_id=7):
        out["typical"] = g.inference_speech_tortoise(refer_t, rl, text_t, do_sample=True, num_return_sequences=1, typical_sampling=True,
                                                     typical_mass=0.9, **kw).numpy()
    for k, v in out.items():
        print(k, v.shape, v.tolist())
    save("gpt_generate_branches", refer=refer, text=text, sample_id=np.array(7), seed=np.array(SEED_N), input_tokens=input_tokens, input_tokens2=input_tokens2, **out)


if __name__ == "__main__":
    main()
