import sys, torch
sys.path.insert(0, "/root/repo")
from audio_generation_amd import ops
IMPL = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # 3 = bf16x3
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = 8
for (cin, cout, kh, kw, sh, sw, h, w) in [(64, 64, 3, 3, 1, 1, 282, 512), (128, 128, 3, 3, 1, 1, 141, 256), (64, 128, 4, 4, 2, 2, 282, 512), (32, 64, 3, 4, 1, 2, 282, 1024)]:
    x = torch.randn(B, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, kh, kw, device="cuda") * 0.05
    bias = torch.randn(cout, device="cuda")
    pad = ((kh - 1) // 2, (kw - 1) // 2)
    d = ops.conv2d_desc(B, cin, cout, h, w, kh, kw, (sh, sw), pad, 0, 0.2, IMPL)
    pk = ops.conv2d_pack(d, wt)
    y = ops.conv2d_forward(d, x, pk, bias)
    dy = torch.randn_like(y)
    pb = ops.conv2d_pack_bwd(d, wt)
    fl = 2.0 * y.numel() * cin * kh * kw
    tf = timeit(lambda: ops.conv2d_forward(d, x, pk, bias))
    tx = timeit(lambda: ops.conv2d_bwd_data(d, dy, pb))
    tw = timeit(lambda: ops.conv2d_bwd_weight(d, x, dy))
    print(f"{cin}->{cout} k{kh}x{kw} s{sh}x{sw} {h}x{w}: fwd {tf:.3f} ms {fl/tf*1e-9:.1f} TF | dx {tx:.3f} ms {fl/tx*1e-9:.1f} TF | dW {tw:.3f} ms {fl/tw*1e-9:.1f} TF   {ops.conv2d_kernel_name(d)}")
