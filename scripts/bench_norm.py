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
for (N, C, hw) in ([] if "--ln-only" in sys.argv else SHAPES):
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

# LayerNorm: rows in flight per wave (1 = the round-1 kernel's schedule); same per-row arithmetic -> bit-identical
for (rows, C) in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (65536, 640), (16384, 1280), (8192, 640), (2048, 1280)]:
    x = torch.randn(rows, C, device="cuda", dtype=torch.float16) * 2 + 0.3
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    res, ref = [], None
    for rpw in (1, 2, 4, 0):
        H.lib().cfgpp_layernorm_set_rows_per_wave(rpw)
        out = H.layernorm(x, g, b)
        if ref is None: ref = out.clone()
        same = bool(torch.equal(out, ref))
        dt = timeit(lambda: H.layernorm(x, g, b))
        res.append(f"rpw{rpw}: {dt*1e6:6.1f} us {4.0*rows*C/dt/1e12:5.2f} TB/s same={same}")
    H.lib().cfgpp_layernorm_set_rows_per_wave(0)
    print(f"ln rows={rows:6d} C={C:4d}  " + "   ".join(res), flush=True)
