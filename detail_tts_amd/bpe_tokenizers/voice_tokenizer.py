"""Text front-end mirror: `VoiceBpeTokenizer` of bpe_tokenizers/voice_tokenizer.py:31-54.

Host-side only: punctuation normalisation, ' ' -> '[SPACE]', then the HuggingFace `tokenizers` BPE model stored in the
reference's vocabulary file (bpe_tokenizers/zh_tokenizer.json etc. — a data asset the caller supplies, like the checkpoint).
The pinyin conversion of api.py:21-22 (pypinyin's dictionary) is not part of this package: pass pinyin text.
"""
from __future__ import annotations

import re

_PUNCT = {"{": "(", "}": ")", "[": "(", "]": ")", "`": "'", "—": "-", "ʼ": "'"}
_PUNCT_RE = re.compile("|".join(re.escape(k) for k in sorted(_PUNCT, key=len, reverse=True)))
_EXTRANEOUS_RE = re.compile(r"^[@#%_=\$\^&\*\+\\]$")


def remove_extraneous_punctuation(word: str) -> str:
    """voice_tokenizer.py:14-28: bracket / quote / dash normalisation; a word that is one stray symbol becomes empty."""
    word = _PUNCT_RE.sub(lambda m: _PUNCT[m.group(0)], word)
    return _EXTRANEOUS_RE.sub("", word)


class VoiceBpeTokenizer:
    def __init__(self, vocab_file):
        self.tokenizer = None
        if vocab_file is not None:
            from tokenizers import Tokenizer
            self.tokenizer = Tokenizer.from_file(str(vocab_file))

    def preprocess_text(self, txt: str) -> str:
        return remove_extraneous_punctuation(txt)

    def encode(self, txt: str):
        txt = self.preprocess_text(txt).replace(" ", "[SPACE]")
        return self.tokenizer.encode(txt).ids

    def decode(self, seq):
        if hasattr(seq, "cpu"):
            seq = seq.cpu().numpy()
        txt = self.tokenizer.decode([int(v) for v in seq], skip_special_tokens=False).replace(" ", "")
        return txt.replace("[SPACE]", " ").replace("[STOP]", "").replace("[UNK]", "")
