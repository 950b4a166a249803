"""vqvae.utils.diffusion (reference: vqvae/utils/diffusion.py) -> detail_tts_amd.vqvae.utils.diffusion"""
from detail_tts_amd.vqvae.utils.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps  # noqa: F401
