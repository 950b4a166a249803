"""bpe_tokenizers.voice_tokenizer (reference: bpe_tokenizers/voice_tokenizer.py:31-54) -> detail_tts_amd mirror"""
from detail_tts_amd.bpe_tokenizers.voice_tokenizer import VoiceBpeTokenizer, remove_extraneous_punctuation  # noqa: F401
