"""vqvae.utils.data_utils (reference: vqvae/utils/data_utils.py:105-185) -> detail_tts_amd mirrors"""
from detail_tts_amd.config import HParams  # noqa: F401
from detail_tts_amd.vqvae.utils.data_utils import Resample, mel_spectrogram_torch, spectrogram_torch  # noqa: F401
