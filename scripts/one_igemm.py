"""Run ONE igemm shape with a forced tile config in a loop (PMC / timing target): python scripts/one_igemm.py conv_l0_640in 5 [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
SH = {"conv_l0_640in": ("conv", 16, 64, 640, 320), "conv_l0_320": ("conv", 16, 64, 320, 320), "conv_l1_640": ("conv", 16, 32, 640, 640),
      "conv_l2_1280": ("conv", 16, 16, 1280, 1280), "lin_m4096_1280": ("lin", 4, 32, 1280, 1280), "lin_m4096_ffout": ("lin", 4, 32, 5120, 1280),
      "geglu_m4096": ("geglu", 4, 32, 1280, 10240), "geglu_l0": ("geglu", 16, 64, 320, 2560),
      "conv_m2048_1280": ("conv", 2, 32, 1280, 1280), "conv_m8192_640": ("conv", 2, 64, 640, 640), "lin_m2048_ffout": ("lin", 2, 32, 5120, 1280),
      "heads_m4096": ("heads", 4, 32, 1280, 3840), "heads_l0": ("heads", 16, 64, 320, 960), "heads_l1": ("heads", 16, 32, 640, 1920)}
kind, R, hw, Cin, N = SH[sys.argv[1]]; cfg = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = R * hw * hw
H.lib().cfgpp_igemm_force_config(cfg)
H.lib().cfgpp_igemm_set_n_major(int(os.environ.get("N_MAJOR", "-1")))
H.lib().cfgpp_igemm_force_split(int(os.environ.get("SPLIT", "0")))
H.lib().cfgpp_igemm_set_staged_epilogue(int(os.environ.get("STAGED", "1")))
if kind == "conv":
    K = 9 * Cin
    x = torch.randn(R, hw + 2, hw + 2, Cin, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5; o = H.empty_pn(R, hw, hw, N)
    fn = lambda: H.igemm(x, None, Cin, 0, 9, 1, hw, hw, w, M, N, out=o, omode=1, old=N)
elif kind == "heads":        # QKV projection into head-major Q / K / V^T (C = K = Cin, heads of 64 or 40)
    from cfgpp_amd import _lib
    K = Cin; C = Cin; d = 40 if C == 320 else 80 if C == 640 else 64; nh = C // d; tokens = hw * hw; dp = H.round_up(d, 32)
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5
    qp, kp = H.round_up(tokens, 128), H.round_up(tokens, 64)
    hq = torch.zeros(R * nh, qp, dp, device="cuda", dtype=torch.float16); hk = torch.zeros(R * nh, kp, dp, device="cuda", dtype=torch.float16)
    hvt = torch.zeros(R * nh, dp, kp, device="cuda", dtype=torch.float16)
    fn = lambda: _lib.check(H.lib().cfgpp_op_igemm_heads(H.P(x), K, H.P(w), M, N, None, tokens, H.P(hq), H.P(hk), H.P(hvt), 0, C, d, nh, qp, kp, H.stream()), "heads")
else:
    K = Cin
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5
    b = torch.zeros(N, device="cuda")
    epi = 1 if kind == "geglu" else 0
    o = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=torch.float16)
    fn = lambda: H.igemm(x, None, K, 0, 1, 0, 0, 0, w, M, N, bias=b, out=o, omode=0, old=o.shape[1], epi=epi)
for _ in range(3): fn()
torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): fn()
e.record(); torch.cuda.synchronize()
dt = s.elapsed_time(e) / iters * 1e-3
print(f"{sys.argv[1]} cfg{cfg} split{os.environ.get('SPLIT', '0')}: {dt*1e6:.1f} us  {2.0*M*N*K/dt/1e12:.0f} TF/s", flush=True)
