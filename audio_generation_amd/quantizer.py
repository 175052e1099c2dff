"""Drop-in residual vector quantiser executed by ``agx_rvq_forward``.

Stands where the reference imports ``ResidualQuantizer`` / ``tuple_checker`` from
the external ``som_quantizer`` module (``networks/vae.py:6``; source absent from
the reference tree -- **parity unpinned**, see ``oracle/rvq_exact.c``).  The
surface below is exactly what the reference's callers touch:

* ctor kwargs of ``vae.py:245-251``;
* ``__call__(x[b l c], codebook_n, update_codebook=, prioritize_early=)`` ->
  ``(x_q, index, commit_loss)`` (``vae.py:315-318``);
* ``.num_quantizers`` (``training.py:183,496``), ``.use_som`` (``utils.py:239``),
  ``.quantizers[i].dequantize(idx)`` (``vae.py:333``),
  ``.quantizers[0].som.height/.width`` (``utils.py:244-245``),
  ``.get_stale_clusters()`` (``training.py:435,461``; ``utils.py:176-181``),
  ``.update_cutoff(new_cutoff=, ratio=)`` (``vae.py:351``), ``.parameters()``
  (``training.py:516``).

The forward search is the HIP kernel; the codebook *update* rules of the
external module (EMA / SOM neighbourhoods) are unknown and training-only, so
``update_codebook=True`` applies a plain EMA k-means update (build-defined,
torch bookkeeping, not on the measured path).

**Not performed** (their definition lives in the absent ``som_quantizer``; parity unpinned): the SOM
neighbourhood update (``use_som`` / ``som_kernel_type``, ``vae.py:249-250``), early-stage prioritisation
(``prioritize_early``, ``training.py:325-328``) and stale-code replacement (``vq_cutoff_freq`` only feeds
``get_stale_clusters()``).  The kwargs are accepted so the reference's constructor calls and shipped YAML
(``config/training.yml``: ``use_som: True``) keep working; the first time one of them would have changed the
result -- ``update_codebook=True`` with ``use_som=True``, or ``prioritize_early=True`` -- a
``UserWarning`` says so, once per process.
"""
from __future__ import annotations

import warnings
from typing import List, Optional

import torch
from torch import nn

from . import ops
from ._lib import AgxError
from .vae import tuple_checker  # re-exported like the external module does (vae.py:6)

Tensor = torch.Tensor
__all__ = ["ResidualQuantizer", "tuple_checker"]


def _near_square(n: int):
    """Factor n into (h, w) as square as possible (utils.py:13-27 semantics)."""
    h = int(n ** 0.5)
    while h > 1 and n % h:
        h -= 1
    return max(h, 1), n // max(h, 1)


class _SomGrid:
    def __init__(self, k: int):
        self.height, self.width = _near_square(k)


class _Stage(nn.Module):
    """View of one residual stage (``quantizer.quantizers[i]``)."""

    def __init__(self, parent: "ResidualQuantizer", i: int):
        super().__init__()
        object.__setattr__(self, "_parent", parent)  # not a submodule: avoids a reference cycle in state_dict
        self.i = i
        self.som = _SomGrid(parent.codebook_size)

    @property
    def codebook(self) -> Tensor:
        return self._parent.codebooks[self.i][:self._parent.codebook_sizes[self.i]]

    def dequantize(self, idx: Tensor) -> Tensor:
        """``(..,) int -> (.., D)`` gather (call site vae.py:333)."""
        return ops.rvq_dequantize(self.codebook.detach(), idx)


_WARNED = set()


def _warn_once(key: str, message: str) -> None:
    if key not in _WARNED:
        _WARNED.add(key)
        warnings.warn(message, UserWarning, stacklevel=3)


class ResidualQuantizer(nn.Module):
    def __init__(self, num_quantizers=8, dim=512, quantizer_class="ema", codebook_sizes=1024,
                 vq_cutoff_freq=1, use_som=True, som_kernel_type="hard", ema_decay=0.99):
        super().__init__()
        sizes = tuple(int(v) for v in tuple_checker(codebook_sizes, num_quantizers))
        if min(sizes) < 1:
            raise ValueError("codebook sizes must be positive")
        self.num_quantizers = int(num_quantizers)
        self.dim = int(dim)
        # One size per stage (vae.py:233).  Storage is (Q, K, D) with K = the largest stage; rows beyond a stage's
        # own size are zero padding that the search kernel can never select (ops.rvq_pack(..., sizes)).
        self.codebook_sizes = sizes
        self.codebook_size = max(sizes)
        self.quantizer_class = quantizer_class
        self.vq_cutoff_freq = vq_cutoff_freq
        self.use_som = use_som
        self.som_kernel_type = som_kernel_type
        self.ema_decay = ema_decay
        init = torch.randn(self.num_quantizers, self.codebook_size, self.dim)
        for q, kq in enumerate(sizes):
            init[q, kq:] = 0.0
        if quantizer_class == "base":          # learnable codebook (config/training.yml: vq_type "base")
            self.codebooks = nn.Parameter(init)
        else:                                   # "ema": statistics-updated, not a parameter
            self.register_buffer("codebooks", init)
        self.register_buffer("cluster_frequency", torch.ones(self.num_quantizers, self.codebook_size))
        self.register_buffer("ema_sum", init.clone())
        self.quantizers = nn.ModuleList([_Stage(self, i) for i in range(self.num_quantizers)])
        self._packed: Optional[Tensor] = None
        self._packed_key = None

    # ------------------------------------------------------------------ kernels
    def _packed_codebooks(self) -> Tensor:
        cb = self.codebooks
        key = (cb.data_ptr(), cb._version)
        if self._packed is None or self._packed_key != key:
            uniform = min(self.codebook_sizes) == self.codebook_size
            self._packed = ops.rvq_pack(cb.detach(), None if uniform else self.codebook_sizes)
            self._packed_key = key
        return self._packed

    def _q_used(self, codebook_n) -> int:
        return self.num_quantizers if codebook_n is None else max(0, min(int(codebook_n), self.num_quantizers))

    def _quantize(self, x: Tensor, layout: str, codebook_n, update_codebook: bool, prioritize_early: bool = False):
        q_used = self._q_used(codebook_n)
        if prioritize_early:
            _warn_once("prioritize_early",
                       "audio_generation_amd.ResidualQuantizer: prioritize_early=True is accepted but IGNORED -- its "
                       "definition lives in the external `som_quantizer` module, which is absent from the reference tree "
                       "(parity unpinned); every stage is treated alike.")
        if update_codebook and self.training and self.use_som:
            _warn_once("use_som",
                       "audio_generation_amd.ResidualQuantizer: use_som=True is accepted but the SOM neighbourhood update "
                       f"(som_kernel_type={self.som_kernel_type!r}) and stale-code replacement are NOT performed -- their "
                       "definition lives in the external `som_quantizer` module, which is absent from the reference tree "
                       "(parity unpinned); update_codebook=True applies the build-defined plain EMA k-means update only.")
        xq, index, sq_err, commit_eval = ops.rvq_forward(x.detach(), self.codebooks.detach(), self._packed_codebooks(),
                                                         q_used, layout)
        if torch.is_grad_enabled() and x.requires_grad:
            # training semantics (build-defined, the external module's are unknown): straight-through
            # estimator for x_q, and a commitment loss that is differentiable in the encoder output --
            # sum over stages of mean((x - sum_{p<=q} c_p)^2) with the selected codewords detached.
            partial, commit = None, x.new_zeros(())
            for q in range(q_used):
                c = ops.rvq_dequantize(self.codebooks.detach()[q], index[..., q])          # (B,T,D)
                c = c if layout == "b l c" else c.transpose(1, 2)
                partial = c if partial is None else partial + c
                commit = commit + ((x - partial) ** 2).mean()
            xq = x + (xq - x).detach()
        else:
            commit = commit_eval                          # sum(sq_err) / numel, written by the search launch itself
        if update_codebook and self.training:
            frames = x if layout == "b l c" else x.transpose(1, 2)
            self._ema_update(frames.reshape(-1, self.dim), index.reshape(-1, q_used))
        return xq, index, commit

    def quantize_bcl(self, x: Tensor, codebook_n=None, update_codebook=False, prioritize_early=False):
        """Native entry used by ``CausalVQAE.encode``: (B,D,T) in, (B,D,T) out."""
        return self._quantize(x, "b c l", codebook_n, update_codebook, prioritize_early)

    def forward(self, x: Tensor, codebook_n=None, update_codebook=False, prioritize_early=False):
        """Reference call contract (vae.py:315-318): (B,T,D) -> (x_q, index, commit_loss)."""
        return self._quantize(x, "b l c", codebook_n, update_codebook, prioritize_early)

    # --------------------------------------------------------- training bookkeeping
    def _invalidate_packed(self) -> None:
        """Drop the packed search image (centred transposed codewords, |c'|^2, cmax, mu).  Every writer of the
        codebooks calls this: writes through ``.data`` do not move ``codebooks._version``."""
        self._packed = None
        self._packed_key = None

    @torch.no_grad()
    def _ema_update(self, frames: Tensor, index: Tensor, stats: Optional[Tensor] = None) -> None:
        """Plain EMA k-means update per stage (build-defined; see module docstring).

        Data parallel (the reference toggles ``update_codebook`` inside the step,
        training.py:305-308, 326): the per-code assignment counts (K,) and residual sums (K,D) of all
        stages travel in ONE all-reduce (sum over ranks), so every replica applies the same global-batch
        statistics and the codebooks stay bit-identical across ranks -- also when the ranks drew different
        ``codebook_n`` (fixed-shape collective, see below).  ``stats`` (q_used, K, D+1): this rank's statistics if the
        caller already holds them (host-logic tests); otherwise one ``agx_rvq_ema_stats`` launch."""
        from . import dist as agx_dist
        q_used = index.shape[1]
        cb = self.codebooks.detach()            # shares storage and version counter with the module's tensor
        # counts and sums of the residual each stage's search saw (against the PRE-update codewords), one launch,
        # added in frame order: the update is reproducible run to run (index_add_ on the device is not)
        own = ops.rvq_ema_stats(frames, cb, index) if stats is None else stats
        # The collective has ONE shape whatever this step's codebook_n: the reference draws codebook_n per process
        # (training.py:294), so ranks may run different numbers of stages -- an all-reduce whose size depends on it
        # would hang or corrupt.  Stages this rank did not run contribute zeros.
        stats = own.new_zeros(self.num_quantizers, self.codebook_size, self.dim + 1)
        stats[:q_used].copy_(own)
        agx_dist.allreduce_sum_(stats)
        for q in range(self.num_quantizers):
            counts, sums = stats[q, :, 0], stats[q, :, 1:]
            # a stage NO rank ran (global count 0) is left untouched, as in the single-process loop over q_used stages;
            # decided on the device from the all-reduced counts: identical on every rank, no host sync
            ran = (counts.sum() > 0).to(counts.dtype)
            decay = 1.0 - ran * (1.0 - self.ema_decay)
            self.cluster_frequency[q].mul_(decay).add_(counts, alpha=1 - self.ema_decay)
            if self.quantizer_class != "base":
                self.ema_sum[q].mul_(decay).add_(sums, alpha=1 - self.ema_decay)
                denom = self.cluster_frequency[q].clamp_min(1e-5).unsqueeze(1)
                cb[q].copy_(torch.where(ran > 0, self.ema_sum[q] / denom, cb[q]))
                cb[q, self.codebook_sizes[q]:] = 0.0
        self._invalidate_packed()

    @torch.no_grad()
    def sync_from_rank0(self) -> None:
        """Broadcast codebooks and EMA statistics from rank 0 (after ``init_from_latents`` on per-rank shards,
        or after loading a checkpoint on one rank)."""
        from . import dist as agx_dist
        agx_dist.broadcast_([self.codebooks.detach(), self.cluster_frequency, self.ema_sum])
        self._invalidate_packed()

    @torch.no_grad()
    def init_from_latents(self, z: Tensor, seed: int = 7, decay: float = 0.6) -> float:
        """Data-driven codebook initialisation (what k-means-style VQ training starts from;
        also how the synthetic benchmark makes the arg-min non-degenerate):
        stage 0 = randomly chosen latent frames + small noise, stage q = randn * sigma * decay^q
        with sigma the per-element std of the frames around their mean.  z is (B, D, T)."""
        frames = z.transpose(1, 2).reshape(-1, self.dim)
        gen = torch.Generator().manual_seed(seed)
        sigma = float((frames - frames.mean(dim=0, keepdim=True)).std())
        pick = torch.randint(0, frames.shape[0], (self.codebook_size,), generator=gen).to(frames.device)
        noise = torch.randn(self.num_quantizers, self.codebook_size, self.dim, generator=gen).to(frames.device)
        cb = self.codebooks.detach()
        cb[0].copy_(frames[pick] + 0.1 * sigma * noise[0])
        for q in range(1, self.num_quantizers):
            cb[q].copy_(noise[q] * (sigma * decay ** q))
        for q, kq in enumerate(self.codebook_sizes):
            cb[q, kq:] = 0.0
        # keep the EMA statistics consistent with the new codewords (codebook = ema_sum / frequency)
        self.cluster_frequency.fill_(1.0)
        self.ema_sum.copy_(cb)
        self._invalidate_packed()
        return sigma

    @torch.no_grad()
    def init_randn(self, sigma: float, seed: int = 7) -> None:
        """SURVEY 8(d)'s synthetic codebooks: ``randn(Q,K,D) * sigma`` from seed 7."""
        gen = torch.Generator().manual_seed(seed)
        cb = self.codebooks.detach()
        cb.copy_((torch.randn(self.num_quantizers, self.codebook_size, self.dim, generator=gen) * sigma).to(cb.device))
        for q, kq in enumerate(self.codebook_sizes):
            cb[q, kq:] = 0.0
        self.cluster_frequency.fill_(1.0)
        self.ema_sum.copy_(cb)
        self._invalidate_packed()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate_packed()

    def get_stale_clusters(self) -> List[int]:
        """Number of codewords per stage whose EMA usage fell below the cutoff."""
        return [int((self.cluster_frequency[q][:self.codebook_sizes[q]] < self.vq_cutoff_freq).sum())
                for q in range(self.num_quantizers)]

    def update_cutoff(self, new_cutoff=None, ratio=None):
        if new_cutoff is not None:
            self.vq_cutoff_freq = new_cutoff
        if ratio is not None:
            self.vq_cutoff_freq = self.vq_cutoff_freq * ratio
