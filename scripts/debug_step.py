import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfgpp_amd import engine as E
from cfgpp_amd.coeffs import ddim_coeffs
from cfgpp_amd.schedule import SchedulerTables
from oracle import sampler as O
g = np.load(os.path.join(ROOT, "tests/golden/sampler_golden.npz"))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
tag, tb, lam = "G2/sd_ddim_cfgpp_h", SchedulerTables(50), 0.6
z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
def run(zi, uc, cc, lam, co, tw, rn):
    zd = zi.clone().cuda(); z0o = torch.empty_like(zd)
    E.step_ddim(zd, z0o, uc.cuda(), cc.cuda(), lam, co, tw, rn)
    return z0o.cpu(), zd.cpu()
i = 0; t = tb.timesteps[i]
zi, uc, cc = z[i][0:1].contiguous(), e[i][0:1].contiguous(), e[i][1:2].contiguous()
co = ddim_coeffs(tb.alpha(t), tb.alpha(int(t) - tb.skip), True)
print("coeffs", co)
a, b = run(zi, uc, cc, lam, co, False, True)
print("full: z0t mism", int((a != z0[i]).sum()), "zt mism", int((b != zt[i]).sum()), "max|dz0|", float((a - z0[i]).abs().max()), "max|dzt|", float((b - zt[i]).abs().max()))
# pure division
a, b = run(zi, uc, cc, lam, (0.0, co[1], 1.0, 0.0), False, True)
print("div only: mism", int((a != zi / torch.tensor(co[1])).sum()), " zn==z0t", int((a != b).sum()))
# pure product uc*c1 (tweedie_uc): z=0 -> z0t = -h(uc*c1)/1
zz = torch.zeros_like(zi)
a, b = run(zz, uc, cc, lam, (co[0], 1.0, 1.0, 0.0), True, True)
ref = -(uc.float() * torch.tensor(co[0])).half().float()
print("prod uc*c1: mism", int((a != ref).sum()), float((a-ref).abs().max()))
# mix: z=0, c1=1 -> z0t = -hat
a, b = run(zz, uc, cc, lam, (1.0, 1.0, 1.0, 0.0), False, True)
hat = O.cfg_mix(uc, cc, lam).float()
print("mix: mism", int((a != -hat).sum()), float((a + hat).abs().max()))
d = (cc.float()-uc.float()).half(); e1 = (d.float()*torch.tensor(0.6)).half(); 
# renoise only: c1=0,c2=1 -> z0t=z ; zn = c3*z + h(c4*uc)
a, b = run(zi, uc, cc, lam, (0.0, 1.0, co[2], co[3]), False, True)
ref = torch.tensor(co[2]) * zi + (uc.float() * torch.tensor(co[3])).half().float()
print("renoise: mism", int((b != ref).sum()), float((b-ref).abs().max()))
idx = (a != -hat).flatten().nonzero().flatten()[:5]
print("examples uc, c, got, want:", [(float(uc.flatten()[k]), float(cc.flatten()[k]), float(a.flatten()[k]), float(-hat.flatten()[k])) for k in idx])
print("---- per step")
for i, t in enumerate(tb.timesteps):
    zi, uc, cc = z[i][0:1].contiguous(), e[i][0:1].contiguous(), e[i][1:2].contiguous()
    co = ddim_coeffs(tb.alpha(t), tb.alpha(int(t) - tb.skip), True)
    a, b = run(zi, uc, cc, lam, co, False, True)
    m0, m1 = int((a != z0[i]).sum()), int((b != zt[i]).sum())
    if m0 or m1:
        a2, b2 = O.ddim_step(zi, uc, cc, lam, tb.alpha(t), tb.alpha(int(t) - tb.skip), False, True)
        k = (a != z0[i]).flatten().nonzero().flatten()[:2]
        print(i, int(t), "z0t", m0, "zt", m1, "oracle-vs-golden", int((a2 != z0[i]).sum()), co,
              [(float(zi.flatten()[j]), float(uc.flatten()[j]), float(cc.flatten()[j]), float(a.flatten()[j]), float(z0[i].flatten()[j])) for j in k])
