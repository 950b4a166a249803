"""Timing of the device prompt front-end at the headline shape: 8 prompts x 10 s at 44.1 kHz -> 24 kHz -> log-mel [8, 128, 937]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detail_tts_amd.runtime import Runtime

rt = Runtime({}, parts=("frontend",), folded=True)
x = torch.randn(8, 441000, device="cuda") * 0.1


def run():
    return rt.mel_spectrogram(rt.resample(x, 44100, 24000))


for _ in range(3):
    m = run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    m = run()
e1.record()
torch.cuda.synchronize()
print(f"resample 44.1k->24k + log-mel, 8 x 10 s: {e0.elapsed_time(e1) / 20:.3f} ms  -> mel {tuple(m.shape)}")
