"""GroupNorm A/B (one-launch slab kernel vs two-launch form) on the UNet's own shapes: python scripts/bench_norm.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
SHAPES = [  # rows, C, hw
    (16, 320, 64), (16, 640, 64), (16, 640, 32), (16, 1280, 32), (16, 1920, 32), (16, 1280, 16), (16, 2560, 16), (16, 1280, 8), (16, 2560, 8),
    (4, 320, 128), (4, 640, 64), (4, 1280, 64), (4, 1280, 32), (4, 2560, 32), (4, 1920, 32), (32, 1280, 32), (32, 640, 64),
]
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e-3
for (N, C, hw) in SHAPES:
    x = torch.randn(N, hw + 2, hw + 2, C, device="cuda", dtype=torch.float16)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    res = []
    ref = None
    for mode in (1, 2):
        H.lib().cfgpp_groupnorm_set_mode(mode)
        out = H.groupnorm(x, None, g, b, 32, 1e-5, 1)
        if ref is None: ref = out.float()
        dev = float((out.float() - ref).abs().max())
        dt = timeit(lambda: H.groupnorm(x, None, g, b, 32, 1e-5, 1))
        res.append(f"mode{mode}: {dt*1e6:6.1f} us {4.0*N*hw*hw*C/dt/1e12:5.2f} TB/s(4B/el) dev {dev:.1e}")
    H.lib().cfgpp_groupnorm_set_mode(0)
    print(f"gn rows={N:2d} C={C:4d} hw={hw:3d}  " + "   ".join(res), flush=True)
