import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cfgpp_amd import engine as E
from mock_engine import emulate_kdiff_denoise
g = torch.Generator().manual_seed(9)
n = (8, 4, 64, 64)
x, y, z = ((torch.randn(n, generator=g) * 2).half() for _ in range(3))
eu, ec = torch.randn(n, generator=g).half(), torch.randn(n, generator=g).half()
dr, ur = torch.empty_like(x), torch.empty_like(x)
emulate_kdiff_denoise(x, eu, ec, 0.6, 3.217, dr, ur)
xd, eud, ecd = x.cuda(), eu.cuda(), ec.cuda()
dd, ud = torch.empty_like(xd), torch.empty_like(xd)
E.kdiff_denoise(xd, eud, ecd, 0.6, 3.217, dd, ud)
torch.cuda.synchronize()
for nm, a, b in (("den", dd.cpu(), dr), ("uden", ud.cpu(), ur)):
    bad = (a != b)
    print(nm, "mismatch", int(bad.sum()), "of", a.numel(), "max|d|", float((a.float()-b.float()).abs().max()))
    idx = bad.flatten().nonzero().flatten()[:4]
    for k in idx:
        k = int(k); print("   x", float(x.flatten()[k]), "uc", float(eu.flatten()[k]), "c", float(ec.flatten()[k]), "gpu", float(a.flatten()[k]), "ref", float(b.flatten()[k]))
