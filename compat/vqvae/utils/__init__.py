"""Reference module path served by detail_tts_amd (see compat/README.md)."""
