"""BASELINE configs at their REAL clip lengths against the CPU oracle.

* config S (``configs[1]``): 4 clips x 72 000 samples, 8 x 1024 x 512 RVQ with data-initialised (non-degenerate)
  codebooks, the whole ``oracle.codec.vqae_forward`` beside the HIP forward: indices bit-exact on identical latents,
  waveform RMS < 1e-4, and the independent path's disagreements PROVED to be near ties
  (``oracle/neartie.py``: both top-2 margins <= 2 |delta| |c_a - c_b| with delta the measured latent difference);
* the same with ENGINEERED near ties in the stage-0 codebook, so the proof is exercised on real flips;
* config 4 (``configs[3]``) at its BASELINE size 8 x 2 x 144 000 with ``wavelet_decoders=[F,T,F,F]`` (the reference's
  wiring, ``vae.py:166-173``): run-to-run determinism, batch independence, one clip against the oracle.

The CPU oracle runs ~0.2 Msamples/s on the box's host share: one 72 000-sample clip is a fraction of a second.
"""
import pytest
import torch

from audio_generation_amd.vae import CausalVQAE
from oracle import codec, neartie, rvq
from tests.helpers import rms

pytestmark = pytest.mark.gpu
DEV = "cuda"
L_S = 72000                     # utils.py:149: the collator's clip length (3 s @ 24 kHz) -> 225 frames (vae.py:354)
LATENT_REL_TOL = 2e-5           # |z_gpu - z_cpu| / |z_cpu| per frame: two fp32 evaluations of a 30-conv stack
WAVE_RMS_TOL = 1e-4             # north_star: reconstructed waveform within 1e-4 RMS


def _config_s(n_clips, seed=1234):
    torch.manual_seed(0)
    kw = dict(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
              codebook_dim=512, input_format="n c l", wavelet_decoders=False)
    model = CausalVQAE(**kw).eval()
    spec = codec.CodecSpec(in_channels=1, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=False, input_format="n c l")
    gen = torch.Generator().manual_seed(seed)
    x = (0.1 * torch.randn(n_clips, 1, L_S, generator=gen)).clamp(-1, 1)      # SURVEY 8(d) inputs
    return model, spec, x


def _check_against_oracle(model, spec, x, sd, min_distinct):
    """The three parity statements; returns the near-tie report of the independent path."""
    cbs = sd["quantizer.codebooks"]
    with torch.no_grad():
        y, commit, index = model(x.to(DEV))
        z_gpu = model._run_encoders(model.rearrange_in(x.to(DEV)))
    b, t = index.shape[:2]
    assert index[..., 0].unique().numel() >= min_distinct, index[..., 0].unique().numel()
    # (1) the oracle's whole forward, independently, on the same input / weights / codebooks
    z_cpu = codec.encode_latents(x, sd, spec)
    zq_cpu, idx_cpu, commit_cpu = rvq.residual_quantize(z_cpu, cbs)
    y_cpu = codec.decode_latents(zq_cpu, sd, spec)
    # (2) bit-exact indices against the definition run on the SAME latents, waveform within the budget
    frames_gpu = z_gpu.cpu().transpose(1, 2).contiguous()
    zq_same, idx_same, commit_same = rvq.residual_quantize(frames_gpu, cbs)
    assert torch.equal(index.cpu(), idx_same)
    assert abs(float(commit) - float(commit_same)) < 1e-5 * max(1.0, float(commit_same))
    assert rms(y.cpu(), codec.decode_latents(zq_same, sd, spec)) < WAVE_RMS_TOL
    # (3) independent path: every first disagreement is a near tie explained by the measured latent difference
    rep = neartie.explain_disagreements(frames_gpu.reshape(b * t, -1).numpy(), z_cpu.reshape(b * t, -1).numpy(),
                                        index.cpu().reshape(b * t, -1).numpy(), idx_cpu.reshape(b * t, -1).numpy(),
                                        cbs.numpy())
    assert rep["max_latent_error_relative"] < LATENT_REL_TOL, rep
    assert rep["proved"], rep
    if rep["frames_with_a_disagreement"] == 0:
        assert rms(y.cpu(), y_cpu) < WAVE_RMS_TOL
        assert abs(float(commit) - float(commit_cpu)) < 1e-5 * max(1.0, float(commit_cpu))
    else:   # clips without a flip must still reproduce the oracle's waveform
        clean = [i for i in range(b) if torch.equal(index[i].cpu(), idx_cpu[i])]
        if clean:
            assert rms(y[clean].cpu(), y_cpu[clean]) < WAVE_RMS_TOL
    return rep


def test_config_s_full_clips_against_the_whole_oracle_forward():
    model, spec, x = _config_s(4)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    z_ref = codec.encode_latents(x[:2], sd, spec)
    model.quantizer.init_from_latents(z_ref.transpose(1, 2))
    sd["quantizer.codebooks"] = model.quantizer.codebooks.detach().clone()
    rep = _check_against_oracle(model.to(DEV), spec, x, sd, min_distinct=64)
    print("config S x 4 clips:", {k: rep[k] for k in ("agreement", "frames_with_a_disagreement", "max_margin_over_bound",
                                                      "max_relative_margin", "max_latent_error_relative")})


def test_config_s_engineered_near_ties_flip_only_within_the_bound():
    """256 stage-0 codewords are placed in pairs z_f +- u around 128 latent frames of the CPU encoder: those frames sit
    on the bisector of their pair up to binary32 rounding, so the HIP encoder's rounding decides many of them the other
    way.  Every such flip must satisfy the margin bound; anything else fails the test."""
    model, spec, x = _config_s(2, seed=4321)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    z_ref = codec.encode_latents(x, sd, spec)                                   # (2, 225, 512)
    sigma = model.quantizer.init_from_latents(z_ref.transpose(1, 2))
    gen = torch.Generator().manual_seed(11)
    frames = z_ref.reshape(-1, 512)
    centre = frames[torch.randperm(frames.shape[0], generator=gen)[:128]]
    u = 0.05 * sigma * torch.randn(128, 512, generator=gen)
    cb = model.quantizer.codebooks.detach()
    cb[0, 0:256:2] = centre + u
    cb[0, 1:256:2] = centre - u
    model.quantizer.ema_sum.copy_(cb)
    model.quantizer._invalidate_packed()
    sd["quantizer.codebooks"] = cb.clone()
    rep = _check_against_oracle(model.to(DEV), spec, x, sd, min_distinct=64)
    print("engineered ties:", {k: rep[k] for k in ("agreement", "frames_with_a_disagreement", "max_margin_over_bound",
                                                   "max_relative_margin", "max_bound_relative")})
    assert rep["frames_with_a_disagreement"] >= 8, rep      # the proof ran on real flips, not on an empty set
    assert rep["max_margin_over_bound"] <= 1.0 + 1e-6, rep  # ... each within the bound its latent difference allows


# ------------------------------------------------------------------------------------------------ config 4
@pytest.fixture(scope="module")
def config4():
    torch.manual_seed(0)
    wd = [False, True, False, False]
    kw = dict(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), num_quantizers=8, codebook_size=1024,
              codebook_dim=512, input_format="n c l", wavelet_decoders=wd)
    model = CausalVQAE(**kw).eval()
    spec = codec.CodecSpec(in_channels=2, n_blocks=4, strides=(2, 4, 5, 8), codebook_dim=512,
                           wavelet_decoders=wd, input_format="n c l")
    gen = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(8, 2, 144000, generator=gen)).clamp(-1, 1)          # 3 s @ 48 kHz stereo, batch 8
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    z0 = codec.encode_latents(x[:1], sd, spec)
    model.quantizer.init_from_latents(z0.transpose(1, 2))
    sd["quantizer.codebooks"] = model.quantizer.codebooks.detach().clone()
    model = model.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        y, commit, index = model(xd)
    return model, spec, sd, x, xd, y, index


def test_config4_full_size_shapes_determinism_and_batch_independence(config4):
    model, spec, sd, x, xd, y, index = config4
    assert tuple(y.shape) == (8, 2, 144000) and tuple(index.shape) == (8, 450, 8)
    assert torch.isfinite(y).all() and index[..., 0].unique().numel() >= 64
    with torch.no_grad():
        y2, _, idx2 = model(xd)
        perm = torch.tensor([3, 0, 7, 5, 1, 6, 2, 4], device=DEV)
        y_p, _, idx_p = model(xd[perm])
        y_s, _, idx_s = model(xd[2:5])
    assert torch.equal(y2, y) and torch.equal(idx2, index)                     # run to run, bit for bit
    assert torch.equal(y_p, y[perm]) and torch.equal(idx_p, index[perm])       # sharding invariant
    assert torch.equal(y_s, y[2:5]) and torch.equal(idx_s, index[2:5])


def test_config4_full_length_clip_against_the_oracle(config4):
    model, spec, sd, x, xd, y, index = config4
    rep = _check_against_oracle(model, spec, x[:1], sd, min_distinct=64)
    print("config 4, clip 0:", {k: rep[k] for k in ("agreement", "frames_with_a_disagreement", "max_margin_over_bound",
                                                    "max_latent_error_relative")})
    with torch.no_grad():                                                       # the batch-8 run gave the same clip
        y1, _, idx1 = model(xd[:1])
    assert torch.equal(y1, y[:1]) and torch.equal(idx1, index[:1])
