#!/usr/bin/env python3
"""Times of the data-path signal ops at batch 32 x 72 000: low-pass biquad, pre-emphasis, Resample 44.1k / 48k -> 24k."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_generation_amd import signal_ops as sg
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = (0.1 * torch.randn(32, 1, 72000, device="cuda")).clamp(-1, 1)
print(f"lowpass_biquad 32 x 72000: {timeit(lambda: sg.lowpass_biquad(x, 24000, 5000.0)):.3f} ms")
print(f"preemphasis    32 x 72000: {timeit(lambda: sg.preemphasis(x, 0.97)):.3f} ms")
for o in (44100, 48000, 16000):
    r = sg.Resample(o, 24000).cuda()
    xi = torch.randn(32, 1, 3 * o, device="cuda")
    print(f"Resample {o} -> 24000, 32 x {3 * o}: {timeit(lambda: r(xi)):.3f} ms (table {tuple(r.kernel.shape)})")
# multi-resolution mel loss of the generator step (training.py:51-78): 7 windows 32 .. 2048 in n_fft = max(w, 512), hop w / 4
wins = tuple(2 ** i for i in range(5, 12))
specs = [sg.MelSpectrogram(24000, max(w, 512), w, w // 4, 64, True).cuda() for w in wins]
y = x.clone().requires_grad_(True)
def mel_fb():
    y.grad = None
    sg.multispectral_reconstruction_loss(x, y, specs, wins).backward()
for w, sp in zip(wins, specs):
    print(f"mel spectrogram window {w:5d}: forward {timeit(lambda: sp(x)):.3f} ms")
print(f"multispectral_reconstruction_loss forward + backward, 32 x 72000: {timeit(mel_fb, 5):.3f} ms")
