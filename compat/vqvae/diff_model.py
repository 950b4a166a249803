"""vqvae.diff_model (reference: vqvae/diff_model.py) -> detail_tts_amd.vqvae.diff_model"""
from detail_tts_amd.vqvae.diff_model import DiffusionTts  # noqa: F401
