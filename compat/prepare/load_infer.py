"""prepare.load_infer (reference: prepare/load_infer.py:8-34) -> detail_tts_amd.prepare.load_infer"""
from detail_tts_amd.prepare.load_infer import load_model  # noqa: F401
