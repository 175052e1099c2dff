import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from audio_generation_amd import ops
from audio_generation_amd._lib import CONV_CAUSAL, IMPL_MFMA, IMPL_MFMA_BF16X3
for (ci, co, k, s, L) in [(512, 512, 3, 1, 225), (64, 128, 9, 4, 4000), (256, 256, 7, 1, 900)]:
    torch.manual_seed(0)
    x = torch.randn(2, ci, L); w = torch.randn(co, ci, k) / (ci * k) ** 0.5
    P = (k - 1) - s + 1
    ref = F.conv1d(F.pad(x.double(), (P, 0)), w.double(), stride=s)
    for impl in (IMPL_MFMA, IMPL_MFMA_BF16X3):
        d = ops.conv_desc(CONV_CAUSAL, 2, ci, co, L, k, s, 1, 0, 0.1, impl)
        y = ops.conv_forward(d, x.cuda(), ops.conv_pack(d, w.cuda()), None).cpu().double()
        n = min(y.shape[-1], ref.shape[-1])
        e = (y[..., :n] - ref[..., :n]).abs()
        print(f"{ci}->{co} K={k} s={s}: impl {impl}: max err / max|y| = {float(e.max()/ref.abs().max()):.2e}  rms err / rms y = {float(e.pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()):.2e}")
