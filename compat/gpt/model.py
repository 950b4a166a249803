"""gpt.model (reference: gpt/model.py) -> detail_tts_amd.gpt.model"""
from detail_tts_amd.gpt.model import UnifiedVoice  # noqa: F401
