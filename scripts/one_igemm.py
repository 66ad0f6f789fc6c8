"""Run ONE igemm shape with a forced tile config in a loop (PMC target): python scripts/one_igemm.py conv_l0_640in 5 [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
SH = {"conv_l0_640in": (16, 64, 640, 320), "conv_l0_320": (16, 64, 320, 320), "conv_l1_640": (16, 32, 640, 640), "conv_l2_1280": (16, 16, 1280, 1280)}
R, hw, Cin, N = SH[sys.argv[1]]; cfg = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
M = R * hw * hw; K = 9 * Cin
x = torch.randn(R, hw + 2, hw + 2, Cin, device="cuda", dtype=torch.float16)
w = torch.randn(N, K, device="cuda", dtype=torch.float16) * K ** -0.5; o = H.empty_pn(R, hw, hw, N)
H.lib().cfgpp_igemm_force_config(cfg)
for _ in range(iters): H.igemm(x, None, Cin, 0, 9, 1, hw, hw, w, M, N, out=o, omode=1, old=N)
torch.cuda.synchronize()
