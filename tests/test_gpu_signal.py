"""Training-loop signal ops against the oracle's restatement of torchaudio (parity unpinned:
torchaudio is absent, see oracle/signal.py)."""
import pytest
import torch

from audio_generation_amd import signal_ops as sg
from oracle import signal as osg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, tol):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= tol * scale + 1e-9, float((a - b).abs().max()) / scale


def test_preemphasis_and_its_gradient():
    x = torch.randn(3, 1, 1000, requires_grad=True)
    y = osg.preemphasis(x, 0.97)
    dy = torch.randn_like(y)
    y.backward(dy)
    xd = x.detach().to(DEV).requires_grad_(True)
    yd = sg.preemphasis(xd, 0.97)
    close(yd, y, 1e-6)
    yd.backward(dy.to(DEV))
    close(xd.grad, x.grad, 1e-6)


def test_lowpass_biquad():
    torch.manual_seed(0)
    x = (0.5 * torch.randn(4, 1, 3000)).clamp(-1, 1)
    for cutoff in (5000.0, 800.0, 11000.0):
        close(sg.lowpass_biquad(x.to(DEV), 24000, cutoff), osg.lowpass_biquad(x, 24000, cutoff), 2e-5)
    big = 3.0 * torch.randn(2, 500)                       # the clamp of lfilter(clamp=True)
    got = sg.lowpass_biquad(big.to(DEV), 24000, 11000.0)
    assert float(got.abs().max()) <= 1.0
    close(got, osg.lowpass_biquad(big, 24000, 11000.0), 2e-5)


def test_mel_filterbank_matches_the_restatement():
    for n_fft in (512, 2048):
        assert torch.equal(sg.melscale_fbanks(n_fft // 2 + 1, 24000, 64), osg.mel_fbanks(n_fft // 2 + 1, 24000, 64))


@pytest.mark.parametrize("window", [32, 64, 128, 256, 512, 1024, 2048])
def test_mel_spectrogram(window):
    torch.manual_seed(window)
    x = 0.2 * torch.randn(2, 1, 6000)
    want = osg.mel_spectrogram(x.squeeze(1), 24000, window).unsqueeze(1)
    spec = sg.MelSpectrogram(sample_rate=24000, n_fft=max(window, 512), win_length=window, hop_length=window // 4,
                             n_mels=64, normalized=True).to(DEV)
    got = spec(x.to(DEV))
    assert got.shape == want.shape == (2, 1, 64, 1 + 6000 // (window // 4))
    close(got, want, 5e-5)


@pytest.mark.parametrize("n_fft,win,hop", [(512, 96, 32), (400, 100, 25), (512, 200, 64), (256, 256, 64), (512, 30, 8), (1024, 1000, 256)])
def test_mel_spectrogram_windows_that_start_inside_a_tap(n_fft, win, hop):
    """The framed DFT keeps only the taps that meet the window (spectral.hip: fdft_geom): windows whose support starts / ends
    inside a hop (left = (n_fft - win) / 2 not a multiple of hop), odd lengths, the full window -- forward against
    torch.stft in float64 (the call torchaudio makes), and the gradient against autograd through it."""
    torch.manual_seed(n_fft + win)
    x = (0.2 * torch.randn(2, 3000)).double().requires_grad_(True)
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    st = torch.stft(x, n_fft, hop, win, window=w, center=True, pad_mode="reflect", normalized=False, return_complex=True)
    fb = osg.mel_fbanks(n_fft // 2 + 1, 24000, 40).double()
    want = torch.einsum("bft,fm->bmt", (st.real ** 2 + st.imag ** 2) / float((w * w).sum()), fb)
    g = torch.randn_like(want)
    want.backward(g)
    spec = sg.MelSpectrogram(24000, n_fft, win, hop, 40, True).to(DEV)
    xd = x.detach().float().to(DEV).requires_grad_(True)
    got = spec(xd)
    close(got, want.detach().float(), 5e-5)
    got.backward(g.float().to(DEV))
    close(xd.grad, x.grad.float(), 5e-5)


def test_multispectral_loss_and_gradient():
    torch.manual_seed(1)
    orig = 0.2 * torch.randn(2, 1, 8000)
    rec = (orig + 0.02 * torch.randn_like(orig)).requires_grad_(True)
    windows = [2 ** i for i in range(5, 12)]
    want = osg.multispectral_reconstruction_loss(orig.squeeze(1), rec.squeeze(1), 24000, windows, spec_loss_weight=0.01)
    want.backward()
    specs = [sg.MelSpectrogram(24000, max(w, 512), w, w // 4, 64, True).to(DEV) for w in windows]
    rec_d = rec.detach().to(DEV).requires_grad_(True)
    got = sg.multispectral_reconstruction_loss(orig.to(DEV), rec_d, specs, windows, spec_loss_weight=0.01)
    assert abs(float(got) - float(want)) <= 2e-4 * abs(float(want))
    got.backward()
    close(rec_d.grad, rec.grad, 2e-3)


def test_full_size_mel_loss_runs():
    """Config-5 shapes: batch 8 x 72 000; finite, positive, differentiable."""
    torch.manual_seed(2)
    orig = (0.1 * torch.randn(8, 1, 72000)).to(DEV)
    rec = (orig + 0.01 * torch.randn_like(orig)).requires_grad_(True)
    windows = [2 ** i for i in range(5, 12)]
    specs = [sg.MelSpectrogram(24000, max(w, 512), w, w // 4, 64, True).to(DEV) for w in windows]
    loss = sg.multispectral_reconstruction_loss(orig, rec, specs, windows)
    loss.backward()
    assert torch.isfinite(loss) and float(loss) > 0 and torch.isfinite(rec.grad).all() and float(rec.grad.abs().max()) > 0


@pytest.mark.parametrize("orig,new,length", [(48000, 24000, 5000), (44100, 24000, 33075), (16000, 24000, 1234),
                                              (22050, 24000, 900), (24000, 24000, 100), (8000, 24000, 7)])
def test_resample(orig, new, length):
    """transforms.Resample (the collator's resampler) against the restatement; parity unpinned (torchaudio absent)."""
    torch.manual_seed(length)
    x = torch.randn(2, 1, length)
    mod = sg.Resample(orig, new).to(DEV)
    got = mod(x.to(DEV))
    want = osg.resample(x, orig, new)
    assert got.shape == want.shape
    close(got, want, 2e-6)
    if orig != new:
        table, width, of, nf = osg.resample_kernel(orig, new)
        assert (mod.of, mod.nf, mod.width) == (of, nf, width)
        assert float((mod.kernel.cpu() - table).abs().max()) < 1e-7


def test_resample_rejects_what_torchaudio_rejects():
    with pytest.raises(ValueError):
        sg.Resample(44100.5, 24000)
    with pytest.raises(ValueError):
        sg.Resample(44100, 24000, lowpass_filter_width=0)


def test_collator_pads_crops_and_resamples_like_the_reference():
    """utils.collator (utils.py:149-175): same torch.randint draws as the reference's loop, so the crops / pads
    land where its own would with the same seed; equal-length clips are dropped (reference quirk)."""
    torch.manual_seed(3)
    clips = [(torch.randn(1, 3000), 0), (torch.randn(1, 500), 1), (torch.randn(1, 1000), 2)]
    rs = sg.Resample(48000, 24000).to(DEV)
    torch.manual_seed(11)
    got = sg.collator(clips, size=1000, resampler=rs)
    torch.manual_seed(11)
    want = []
    for x, _ in clips:                          # the reference's loop on the oracle's resampler
        x = osg.resample(x, 48000, 24000)
        n = x.shape[-1]
        if n < 1000:
            split = torch.randint(0, 1000 - n, (1,)).item()
            want.append(torch.cat([torch.zeros(1, split), x, torch.zeros(1, 1000 - n - split)], dim=-1))
        elif n > 1000:
            start = torch.randint(0, n - 1000, (1,)).item()
            want.append(x[:, start:start + 1000])
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        assert g.shape == (1, 1000) and not g.is_cuda
        close(g, w, 2e-6)
    assert sg.collator([(torch.zeros(1, 1000), 0)], size=1000) == []
