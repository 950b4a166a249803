"""vqvae.model_24k (reference: vqvae/model_24k.py) -> detail_tts_amd.vqvae.model_24k"""
from detail_tts_amd.vqvae.model_24k import (Generator, SynthesizerTrn, denormalize_torch_mel, do_spectrogram_diffusion,  # noqa: F401
                                            normalize_torch_mel, write_wav)
