"""Adversarial operands for the RVQ's bf16-pipe score GEMM.

The fast path (csrc/rvq.hip) scores codewords on ``v_mfma_f32_32x32x16_bf16`` from two bf16 pieces per operand and
keeps every codeword whose score interval reaches the minimum; the DEFINING binary64 distance decides among them.
The interval's accumulation term rests on a MODEL of the instruction (addends aligned to the largest and truncated:
35 x 2^-24 per instruction).  If some operand pattern broke the model the kernel would silently drop the true
arg-min.  These cases feed the patterns a floating-point adder is weakest on -- per-dimension magnitudes spread over
2^20, sign-alternating cancellation, one huge dimension, pieces near the bf16 subnormal range, the widest frame the
kernel takes (D = 560) -- with planted near-tie codeword pairs, and demand indices == the full defining search
(``oracle.rvq.exact_search``, plain C) and a bit-identical x_q.  Every case also runs with the debug knob
``rvq_verify`` (the defining search on the device beside the fast path): 0 mismatches.

Reference contract: ``quantizer(x, codebook_n, ...) -> (x_q, index, loss)`` (vae.py:315-318); the arithmetic that
defines "exact" is oracle/rvq_exact.c (parity unpinned vs the absent ``som_quantizer``).
"""
import ctypes

import numpy as np
import pytest
import torch

from audio_generation_amd import _lib, ops
from oracle import rvq

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None


def _exact_reference(x: torch.Tensor, cbs: torch.Tensor):
    """The definition, stage by stage: full exact search, r <- r - c (binary32), out <- out + c (binary32)."""
    b, t, d = x.shape
    r = x.reshape(-1, d).numpy().astype(np.float32).copy()
    out = np.zeros_like(r)
    idx = np.empty((r.shape[0], cbs.shape[0]), np.int64)
    for q in range(cbs.shape[0]):
        cb = cbs[q].numpy().astype(np.float32)
        i = rvq.exact_search(r, cb)
        idx[:, q] = i
        r = (r - cb[i]).astype(np.float32)
        out = (out + cb[i]).astype(np.float32)
    return torch.from_numpy(out).reshape(b, t, d), torch.from_numpy(idx).reshape(b, t, -1)


def _verify_counts(reset=False):
    out = (ctypes.c_int64 * 3)()
    _lib.check(_lib.load().agx_rvq_verify_counts(out, 1 if reset else 0), "agx_rvq_verify_counts")
    return [int(v) for v in out]


def _run(x, cbs):
    lib = _lib.load()
    want_q, want_i = _exact_reference(x, cbs)
    xd, cd = x.to(DEV), cbs.to(DEV)
    packed = ops.rvq_pack(cd)
    _verify_counts(reset=True)
    lib.agx_set_tuning(b"rvq_verify", 1)
    try:
        xq, idx, _, _ = ops.rvq_forward(xd, cd, packed, cbs.shape[0])
        torch.cuda.synchronize()
    finally:
        lib.agx_set_tuning(b"rvq_verify", 0)
    bad_codes, bad_frames, checked = _verify_counts(reset=True)
    n_bad = int((idx.cpu() != want_i).sum())
    assert n_bad == 0, f"{n_bad} of {want_i.numel()} indices differ from the full defining search"
    assert torch.equal(xq.cpu(), want_q), "x_q is not bit-identical"
    assert checked == want_i.numel() and bad_codes == 0 and bad_frames == 0, (bad_codes, bad_frames, checked)
    return want_i


def _plant_near_ties(cbs, gen, rel):
    """Every odd codeword = its even neighbour + a perturbation ``rel`` x smaller: the top-2 margin of most frames is
    far inside the score interval, so the candidate rule (not luck) has to keep the right one."""
    q, k, d = cbs.shape
    pert = torch.randn(q, k // 2, d, generator=gen) * rel
    cbs[:, 1::2] = cbs[:, 0:2 * (k // 2):2] * (1.0 + pert)
    return cbs


@pytest.mark.parametrize("d", [512, 560])
def test_magnitudes_spread_over_2_pow_20(d):
    """Dimension d carries magnitude 2^-e_d, e_d uniform in [0, 20] (shuffled): inside every 16-deep MFMA block the
    addends differ by up to 2^40."""
    gen = torch.Generator().manual_seed(100 + d)
    scale = torch.pow(2.0, -20.0 * torch.rand(d, generator=gen))
    q, k = 3, 1024
    cbs = torch.randn(q, k, d, generator=gen) * scale
    cbs[1:] *= 0.5
    _plant_near_ties(cbs, gen, 2.0 ** -12)
    x = torch.randn(2, 48, d, generator=gen) * scale
    x[0, :24] = cbs[0, torch.randint(0, k, (24,), generator=gen)] * (1 + 2.0 ** -10 * torch.randn(24, d, generator=gen))
    _run(x, cbs)


@pytest.mark.parametrize("d", [512, 560])
def test_sign_alternating_cancellation(d):
    """r'.c' = a sum of large products of alternating sign that cancels to ~2^-10 of sum |a b| -- the accumulation
    error is relative to the sum of magnitudes, the score that decides is the small remainder."""
    gen = torch.Generator().manual_seed(200 + d)
    alt = torch.tensor([1.0, -1.0]).repeat((d + 1) // 2)[:d]
    q, k = 2, 512
    base = 1.0 + 2.0 ** -9 * torch.randn(q, k, d, generator=gen)
    cbs = base * torch.ones(d)                        # codewords ~ (+1, +1, +1, ...) (1 + small)
    cbs[:, ::3] *= -1.0                               # a third of the codebook mirrored: the mean codeword stays small
    _plant_near_ties(cbs, gen, 2.0 ** -14)
    x = (alt * (1.0 + 2.0 ** -9 * torch.randn(2, 40, d, generator=gen)))       # frames ~ (+1, -1, +1, ...)
    _run(x, cbs)


@pytest.mark.parametrize("d", [512, 560])
def test_one_huge_dimension(d):
    """One dimension of magnitude 2^10 next to 2^-6 everywhere else: its product (2^20) truncates the other 15 addends of
    its MFMA block at 2^-3 of their size; the small dimensions decide between the planted pairs."""
    gen = torch.Generator().manual_seed(300 + d)
    q, k = 2, 1024
    cbs = torch.randn(q, k, d, generator=gen) * 2.0 ** -6
    big = 37 % d
    cbs[:, :, big] = torch.randn(q, k, generator=gen).sign() * 1024.0 * (1 + 0.01 * torch.randn(q, k, generator=gen))
    cbs[:, 1::2] = cbs[:, 0::2]                                         # pairs equal in the huge dimension ...
    cbs[:, 1::2, :big] += 2.0 ** -12 * torch.randn(q, k // 2, big, generator=gen)    # ... apart in the small ones
    x = torch.randn(2, 48, d, generator=gen) * 2.0 ** -6
    x[..., big] = torch.randn(2, 48, generator=gen).sign() * 1024.0
    _run(x, cbs)


def test_pieces_near_the_bf16_subnormal_range():
    """Operands of magnitude 2^-62 .. 2^-64: the middle bf16 piece of every element (2^-8 below it) and the products
    (2^-124 .. 2^-128) sit at the bottom of the bf16 / fp32 normal range, |c'|^2 underflows towards the fp32 subnormals.
    Whatever the matrix pipe flushes there, the interval must still contain the true arg-min (here most frames end in
    the binary64 decision or the full defining search)."""
    gen = torch.Generator().manual_seed(400)
    q, k, d = 2, 256, 512
    s = 2.0 ** -62
    cbs = torch.randn(q, k, d, generator=gen) * s
    _plant_near_ties(cbs, gen, 2.0 ** -10)
    x = torch.randn(1, 40, d, generator=gen) * s * 0.5
    _run(x, cbs)
    # mixed: ordinary latents against a codebook whose MIDDLE pieces are bf16 subnormals (elements ~2^-118, pieces ~2^-126),
    # and the other way round
    t = 2.0 ** -118
    tiny = torch.randn(q, k, d, generator=gen) * t
    _run(torch.randn(1, 33, d, generator=gen), tiny)
    _run(torch.randn(1, 33, d, generator=gen) * t, torch.randn(q, k, d, generator=gen))
    _run(torch.randn(1, 33, d, generator=gen), cbs)


def test_large_common_offset_with_tiny_spread_d560():
    """The centring step's worst case at the widest frame: offset 2^6, spread 2^-8, K = 1024, planted pairs."""
    gen = torch.Generator().manual_seed(500)
    q, k, d = 3, 1024, 560
    mean = 64.0 * torch.randn(d, generator=gen)
    cbs = torch.randn(q, k, d, generator=gen) * 2.0 ** -8
    cbs[0] += mean
    cbs[1:] *= 0.5
    cbs[:, 1::2] = cbs[:, 0::2] + 2.0 ** -16 * torch.randn(q, k // 2, d, generator=gen)
    x = mean + torch.randn(2, 64, d, generator=gen) * 2.0 ** -8
    want_i = _run(x, cbs)
    assert want_i[..., 0].unique().numel() > 30


def test_mfma_rounding_model_at_instruction_level():
    """The model the accumulation term rests on, measured on THIS device: tools/mfma_bf16_err.hip (built by
    ``__graft_entry__.build()``) runs ``v_mfma_f32_32x32x16_bf16`` on nine operand families -- the five of round 3 plus one
    product 2^20 above fifteen same-signed ones, a huge +/- pair that cancels, products at the bottom of the fp32 range,
    exponents spread over 2^40 against an accumulator of the other sign -- and compares every result with the exact sum: the
    worst error must stay inside the 35 x 2^-24 (|c| + sum |a b|) the kernel allows (measured: 7.05)."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "mfma_bf16_err_bin")
    if not os.path.exists(exe):          # normally built by __graft_entry__.build(); the GPU box has the same hipcc
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-w",
                               os.path.join(root, "tools", "mfma_bf16_err.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    m = re.search(r"WORST_UNITS ([0-9.]+) ALLOWED 35 (\w+)", out.stdout)
    assert m, out.stdout[-500:] + out.stderr[-500:]
    print(out.stdout)
    assert out.returncode == 0 and m.group(2) == "OK" and float(m.group(1)) <= 35.0, out.stdout


def test_verify_mode_catches_a_wrong_index():
    """The checker itself: feed agx_rvq_forward's verify pass a codebook tensor that differs from the packed search
    image in ONE codeword (the fast path searches the stale image): the counters must report the frames that moved."""
    gen = torch.Generator().manual_seed(600)
    q, k, d = 1, 64, 64
    cbs = torch.randn(q, k, d, generator=gen)
    x = torch.randn(1, 64, d, generator=gen)
    x[0, :8] = cbs[0, 5] + 0.01 * torch.randn(8, d, generator=gen)      # eight frames next to codeword 5
    cd = cbs.to(DEV)
    packed = ops.rvq_pack(cd)
    moved = cd.clone()
    moved[0, 5] += 10.0                                                 # the definition now sees codeword 5 far away
    lib = _lib.load()
    _verify_counts(reset=True)
    lib.agx_set_tuning(b"rvq_verify", 1)
    try:
        _, idx, _, _ = ops.rvq_forward(x.to(DEV), moved, packed, 1)     # stale image + moved codebook
        torch.cuda.synchronize()
    finally:
        lib.agx_set_tuning(b"rvq_verify", 0)
    bad_codes, bad_frames, checked = _verify_counts(reset=True)
    assert checked == 64 and bad_codes == bad_frames and bad_codes >= 8, (bad_codes, bad_frames, checked)
    assert int((idx[0, :8, 0] == 5).sum()) == 8
