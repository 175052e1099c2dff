"""oracle/signal.py (parity unpinned vs torchaudio, which is absent): what CAN be pinned is pinned here --
the spectrogram against torch.stft (the call torchaudio's Spectrogram makes), the filterbank's defining
properties, the biquad against scipy's lfilter with the RBJ coefficients."""
import math

import numpy as np
import pytest
import torch

from oracle import signal as osg


@pytest.mark.parametrize("window", [32, 256, 2048])
def test_spectrogram_part_equals_torch_stft(window):
    torch.manual_seed(window)
    x = torch.randn(2, 5000)
    n_fft, hop = max(window, 512), window // 4
    win = torch.hann_window(window, periodic=True)
    spec = torch.stft(x, n_fft, hop, window, win, center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)
    power = (spec / win.pow(2).sum().sqrt()).abs().pow(2)                     # torchaudio: normalized="window", power=2
    want = torch.matmul(power.transpose(-1, -2), osg.mel_fbanks(n_fft // 2 + 1, 24000, 64)).transpose(-1, -2)
    got = osg.mel_spectrogram(x, 24000, window)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_filterbank_properties():
    fb = osg.mel_fbanks(257, 24000, 64)
    assert fb.shape == (257, 64) and float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0 + 1e-6
    peaks = fb.argmax(dim=0)
    assert torch.all(peaks[1:] >= peaks[:-1])                                  # triangles march up in frequency
    # HTK mel scale: centres equally spaced in mel
    hz = torch.linspace(0, 12000, 257)[peaks].double()
    mel = 2595.0 * torch.log10(1.0 + hz / 700.0)
    step = (2595.0 * math.log10(1 + 12000 / 700.0)) / 65
    assert float((mel[8:] - step * torch.arange(9, 65)).abs().max()) < step   # within one bin of the ideal centre


def test_biquad_equals_scipy_lfilter():
    from scipy.signal import lfilter
    torch.manual_seed(0)
    x = (0.3 * torch.randn(3, 2000))
    w0 = 2 * math.pi * 5000 / 24000
    alpha = math.sin(w0) / 2 / 0.707
    b = np.array([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2]) / (1 + alpha)
    a = np.array([1 + alpha, -2 * math.cos(w0), 1 - alpha]) / (1 + alpha)
    want = np.clip(lfilter(b, a, x.double().numpy(), axis=-1), -1, 1)
    got = osg.lowpass_biquad(x, 24000, 5000.0).double().numpy()
    assert np.abs(got - want).max() < 2e-5


def test_preemphasis_definition():
    x = torch.arange(6.0).reshape(1, 6)
    assert torch.allclose(osg.preemphasis(x, 0.5), torch.tensor([[0.0, 1.0, 1.5, 2.0, 2.5, 3.0]]))


@pytest.mark.parametrize("orig,new", [(48000, 24000), (44100, 24000), (16000, 24000), (22050, 24000)])
def test_resample_restatement(orig, new):
    """Parity unpinned (torchaudio absent).  The conv form against the interpolation written sample by sample in
    float64, y[i] = sum_j x[j] h(i / nf - j / of), and a band-limited sine against its analytic resampling."""
    torch.manual_seed(orig % 97)
    x = torch.randn(2, 700)
    y = osg.resample(x, orig, new)
    table, width, of, nf = osg.resample_kernel(orig, new)
    assert y.shape == (2, math.ceil(700 * nf / of))
    base, lpw = min(of, nf) * 0.99, 6
    for i in [0, 1, 5, y.shape[1] // 2, y.shape[1] - 2, y.shape[1] - 1]:
        acc = 0.0
        for j in range(700):
            t = base * (j / of - i / nf)
            if abs(t) >= lpw:
                continue                      # the Hann window is zero from there on
            h = (1.0 if t == 0 else math.sin(math.pi * t) / (math.pi * t)) * math.cos(t * math.pi / lpw / 2) ** 2
            acc += float(x[1, j]) * h * base / of
        assert abs(acc - float(y[1, i])) < 2e-5, (i, acc, float(y[1, i]))
    n = torch.arange(4000, dtype=torch.float64)
    f0 = 0.05 * min(orig, new)                # well inside both bands
    got = osg.resample(torch.sin(2 * math.pi * f0 * n / orig).float()[None], orig, new)[0]
    m = torch.arange(got.shape[0], dtype=torch.float64)
    want = torch.sin(2 * math.pi * f0 * m / new)
    assert float((got.double() - want)[200:-200].abs().max()) < 5e-3
