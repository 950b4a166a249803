"""Configuration of the detail_tts acoustic-synthesis hot path.

The values mirror the reference's single config file
(/root/reference/vqvae/configs/config_24k.json) for the three blocks the
inference path reads: ``data``, ``diffusion``, ``gpt`` and ``vaegan``.  The
stray ``"g_channels"`` key of the reference's ``diffusion`` block
(config_24k.json:60) is accepted and ignored, because the reference's own
``DiffusionTts.__init__`` (vqvae/diff_model.py:134-148) has no such argument.

``HParams`` reproduces the attribute/dict access semantics of the reference's
config object (vqvae/utils/data_utils.py:157-185) so code written against the
reference (``hps.data.hop_length`` / ``hps['data']``) keeps working.
"""
from __future__ import annotations

import copy
import json
import os

# Hot-path constants the reference hard-codes in source rather than in json.
TRAINED_DIFFUSION_STEPS = 4000      # vqvae/model_24k.py:558
INFER_DIFFUSION_STEPS = 50          # vqvae/model_24k.py:581
COND_FREE_K = 2.0                   # vqvae/model_24k.py:562
MEL_MIN = -11.512925465             # vqvae/model_24k.py:501
TORCH_MEL_MAX = 2.7                 # vqvae/model_24k.py:503
NOISE_SCALE = 0.667                 # vqvae/model_24k.py:848
LRELU_SLOPE = 0.1                   # vqvae/modules/modules.py:13
MAX_GENERATE_LENGTH = 600           # vqvae/model_24k.py:792
TOP_P = 0.8                         # vqvae/model_24k.py:787
TEMPERATURE = 0.8                   # vqvae/model_24k.py:788
REPETITION_PENALTY = 2.0            # vqvae/model_24k.py:791

DEFAULT_CONFIG = {
    "data": {
        "sampling_rate": 24000,
        "filter_length": 1024,
        "hop_length": 256,
        "win_length": 1024,
        "n_mel_channels": 128,
        "mel_fmin": 0.0,
        "mel_fmax": None,
    },
    "train": {"segment_size": 10240, "target": "gpt", "mel_weight": 1, "text_weight": 0.01},
    "diffusion": {
        "model_channels": 768,
        "num_layers": 10,
        "in_channels": 128,
        "out_channels": 256,
        "in_latent_channels": 768,
        "in_tokens": 8193,
        "dropout": 0,
        "use_fp16": False,
        "num_heads": 16,
        "layer_drop": 0.2,
        "unconditioned_percentage": 0.15,
    },
    "gpt": {
        "model_dim": 768,
        "max_mel_tokens": 1600,
        "max_text_tokens": 800,
        "heads": 16,
        "mel_length_compression": 1024,
        "use_mel_codes_as_input": True,
        "layers": 10,
        "number_text_tokens": 256,
        "number_mel_codes": 8194,
        "start_mel_token": 8192,
        "stop_mel_token": 8193,
        "start_text_token": 255,
        "train_solo_embeddings": False,
        "spec_channels": 128,
    },
    "vaegan": {
        "inter_channels": 192,
        "hidden_channels": 192,
        "filter_channels": 512,
        "vq_bins": 8192,
        "n_heads": 4,
        "n_layers": 3,
        "kernel_size": 3,
        "p_dropout": 0.1,
        "resblock": "1",
        "resblock_kernel_sizes": [3, 7, 11],
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "upsample_rates": [8, 4, 2, 2, 2],
        "upsample_initial_channel": 400,
        "upsample_kernel_sizes": [16, 8, 2, 2, 2],
        "gin_channels": 768,
    },
}


class HParams:
    """Attribute + mapping access over a (nested) dict, like the reference's."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, dict):
                v = HParams(**v)
            self.__dict__[k] = v

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return self.__dict__[key]

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return repr(self.to_dict())

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in self.__dict__.items()}


def load_config(path_or_dict=None) -> dict:
    """Return a plain nested dict with every key the hot path needs.

    ``path_or_dict`` may be None (built-in defaults == config_24k.json), a path
    to a json file in the reference's format, or an already-parsed dict.
    Missing blocks/keys fall back to the defaults; ``diffusion.g_channels`` is
    dropped (see module docstring).
    """
    cfg = copy.deepcopy(DEFAULT_CONFIG)
    if path_or_dict is None:
        return cfg
    if isinstance(path_or_dict, (str, os.PathLike)):
        with open(os.path.expanduser(path_or_dict)) as f:
            user = json.load(f)
    elif isinstance(path_or_dict, HParams):
        user = path_or_dict.to_dict()
    else:
        user = dict(path_or_dict)
    for block, vals in user.items():
        if isinstance(vals, dict):
            cfg.setdefault(block, {}).update(vals)
        else:
            cfg[block] = vals
    cfg["diffusion"].pop("g_channels", None)
    return cfg
