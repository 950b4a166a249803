"""load_model mirror (reference: prepare/load_infer.py:8-34)."""
from __future__ import annotations

import os

from ..config import load_config
from ..vqvae.model_24k import SynthesizerTrn


def load_model(model_name, model_path, config_path, device):
    """Same signature as the reference.  `model_path` is a torch checkpoint holding the state dict under 'G' or 'model'
    (train.py:139-150), or the string 'synthetic[:SEED]' for deterministic random-init weights (no checkpoint ships
    with the reference repo).  Only model_name == 'vqvae' (the inference model) is supported."""
    if model_name != "vqvae":
        raise NotImplementedError("only the 'vqvae' (SynthesizerTrn) inference model is on the hot path")
    cfg = load_config(os.path.expanduser(config_path) if isinstance(config_path, str) else config_path)
    if isinstance(model_path, str) and model_path.startswith("synthetic"):
        from ..weights import synthetic_state_dict
        seed = int(model_path.split(":")[1]) if ":" in model_path else 0
        sd = synthetic_state_dict(seed, cfg)
    else:
        import torch
        ck = torch.load(os.path.expanduser(model_path), map_location="cpu")
        sd = ck.get("G", ck.get("model", ck))
    return SynthesizerTrn(sd, cfg, device=device).eval()
