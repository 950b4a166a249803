#!/usr/bin/env python3
"""The reference's api.py flow (api.py:10-50) on the MI355X path: reference wav -> resample -> log-mel -> GPT codes -> diffusion ->
vocoder -> gen.wav.  Everything after reading the file runs in libdetail_hip.so.

    python examples/api.py --ckpt /path/model-480.pt --vocab /path/bpe_tokenizers/zh_tokenizer.json --wav 1.wav \\
        --text "da4 jia1 hao3 ， jin1 tian1 lai2 dian3 da4 jia1 xiang3 kan4 de5 dong1 xi1 。"
    python examples/api.py --synthetic            # no checkpoint / vocabulary: seed-0 random weights, random ids, a synthetic prompt
"""
import argparse
import os
import sys
import wave

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detail_tts_amd.prepare.load_infer import load_model                                    # api.py:27
from detail_tts_amd.vqvae.model_24k import write_wav
from detail_tts_amd.vqvae.utils.data_utils import HParams, Resample, load_config, mel_spectrogram_torch   # api.py:29


def read_wav(path):
    with wave.open(path, "rb") as f:
        sr, n, ch, sw = f.getframerate(), f.getnframes(), f.getnchannels(), f.getsampwidth()
        pcm = np.frombuffer(f.readframes(n), {1: np.uint8, 2: np.int16, 4: np.int32}[sw]).reshape(-1, ch)
    scale = {1: 128.0, 2: 32768.0, 4: 2147483648.0}[sw]
    x = (pcm.astype(np.float32) - (128.0 if sw == 1 else 0.0)) / scale
    return torch.from_numpy(x.T[:1].copy()), sr                                           # first channel (api.py:36-37)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default="synthetic:0")
    ap.add_argument("--vocab")
    ap.add_argument("--wav")
    ap.add_argument("--text", "--pinyin", dest="text", default="da4 jia1 hao3",
                    help="the sentence AS PINYIN (pypinyin Style.TONE3, neutral tone 5, space separated: what api.py:21 produces from "
                         "Chinese characters; pypinyin's dictionary is not available offline, so the conversion is the caller's)")
    ap.add_argument("--print-ids", action="store_true", help="print the text token ids as JSON and exit (no GPU needed)")
    ap.add_argument("--ids", help="comma-separated text token ids (bypasses pypinyin + tokenizer: e.g. the demo.ipynb KAT ids)")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--out", default="gen.wav")
    ap.add_argument("--max-generate-length", type=int, default=600)
    a = ap.parse_args()
    device = "cuda:0"
    if a.ids:
        ids = [int(v) for v in a.ids.split(",")]
    elif a.vocab:
        from detail_tts_amd.bpe_tokenizers.voice_tokenizer import VoiceBpeTokenizer
        ids = VoiceBpeTokenizer(a.vocab).encode(" " + a.text.strip() + " ")                 # api.py:22-24
    else:
        ids = np.random.RandomState(0).randint(3, 255, 24).tolist()
    if a.print_ids:
        import json
        print(json.dumps(ids))
        return
    text_tokens = F.pad(torch.IntTensor(ids).unsqueeze(0), (0, 1))                          # api.py:24-25
    vqvae = load_model("vqvae", a.ckpt, None, device)                                       # api.py:33
    if a.wav:
        audio, sr = read_wav(a.wav)                                                         # api.py:34-37
    else:
        sr = 44100
        t = torch.arange(3 * sr) / sr
        audio = (0.2 * torch.sin(2 * np.pi * 220 * t) * torch.exp(-t) + 0.02 * torch.randn(3 * sr))[None]
    audio = Resample(sr, 24000, rt=vqvae.rt)(audio)                                         # api.py:39
    hps = HParams(**load_config())
    spec = mel_spectrogram_torch(audio, hps.data.filter_length, hps.data.n_mel_channels, hps.data.sampling_rate, hps.data.hop_length,
                                 hps.data.win_length, hps.data.mel_fmin, hps.data.mel_fmax, rt=vqvae.rt)      # api.py:41-47
    spec_lengths = torch.LongTensor([spec.shape[-1]])
    text_lengths = torch.LongTensor([text_tokens.shape[-1]])
    with torch.no_grad():
        wav = vqvae.infer(text_tokens, text_lengths, spec, spec_lengths, max_generate_length=a.max_generate_length, seed=a.seed,
                          suppress_eos=a.synthetic or a.ckpt.startswith("synthetic"))       # api.py:49
    write_wav(a.out, wav.squeeze(0), 24000)                                                  # api.py:50
    print(f"{a.out}: {wav.shape[-1] / 24000:.2f} s of audio from {spec.shape[-1]} prompt frames and {len(ids)} text ids")


if __name__ == "__main__":
    main()
