"""A/B micro-benchmark of the implicit-GEMM kernel on the SD1.5 / SDXL shapes.
   python scripts/bench_igemm.py [shape-filter] ; env CFGS="1,2,3,21,22" ITERS=20"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
SHAPES = [  # name, kind, batch rows, HW side, Cin, Cout(N)
    ("conv_l0_320", "conv", 16, 64, 320, 320), ("conv_l0_640in", "conv", 16, 64, 640, 320), ("conv_l1_640", "conv", 16, 32, 640, 640),
    ("conv_l2_1280", "conv", 16, 16, 1280, 1280), ("conv_l3_1280", "conv", 16, 8, 1280, 1280), ("conv_l3_2560in", "conv", 16, 8, 2560, 1280),
    ("xl_conv_64_640", "conv", 4, 64, 640, 640), ("xl_conv_32_1280", "conv", 4, 32, 1280, 1280),
    ("lin_l0_320x320", "lin", 16, 64, 320, 320), ("lin_l0_qkv", "lin", 16, 64, 320, 960), ("lin_l0_geglu", "lin", 16, 64, 320, 2560),
    ("lin_l0_ffout", "lin", 16, 64, 1280, 320), ("lin_l1_geglu", "lin", 16, 32, 640, 5120), ("lin_l2_geglu", "lin", 16, 16, 1280, 10240),
    ("xl_geglu_32", "lin", 4, 32, 1280, 10240), ("xl_ffout_32", "lin", 4, 32, 5120, 1280), ("xl_qkv_32", "lin", 4, 32, 1280, 3840),
    ("xl_lin_32_1280", "lin", 4, 32, 1280, 1280), ("xl_lin_64_640", "lin", 4, 64, 640, 640), ("xl_ffout_64", "lin", 4, 64, 2560, 640),
    ("lin_l1_640x640", "lin", 16, 32, 640, 640), ("lin_l1_ffout", "lin", 16, 32, 2560, 640), ("lin_l1_qkv", "lin", 16, 32, 640, 1920),
    ("lin_l2_1280", "lin", 16, 16, 1280, 1280), ("lin_l2_ffout", "lin", 16, 16, 5120, 1280), ("lin_l2_qkv", "lin", 16, 16, 1280, 3840),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,4,5,6,7,8,10").split(",")]
iters = int(os.environ.get("ITERS", "20"))
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e-3
for name, kind, R, hw, Cin, N in SHAPES:
    if flt and flt not in name: continue
    M = R * hw * hw
    if kind == "conv":
        x = torch.randn(R, hw + 2, hw + 2, Cin, device="cuda", dtype=torch.float16); K = 9 * Cin
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5; o = H.empty_pn(R, hw, hw, N)
        fn = lambda: H.igemm(x, None, Cin, 0, 9, 1, hw, hw, w, M, N, out=o, omode=1, old=N)
    else:
        K = Cin; x = torch.randn(M, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5; o = torch.empty(M, N, device="cuda", dtype=torch.float16)
        fn = lambda: H.igemm(x, None, K, 0, 1, 0, 0, 0, w, M, N, out=o, omode=0, old=N)
    res = []
    H.lib().cfgpp_igemm_force_config(1); fn(); ref = o.float().clone()
    for c in cfgs:
        H.lib().cfgpp_igemm_force_config(c)
        try:
            dt = timeit(fn); bad = "" if float((o.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) else "!WRONG"
            res.append(f"cfg{c}:{2.0*M*N*K/dt/1e12:6.0f}{bad}")
        except Exception as e: res.append(f"cfg{c}: ERR")
    H.lib().cfgpp_igemm_force_config(0)
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d}  " + "  ".join(res), flush=True)
