"""How does torch-ROCm combine a 0-dim DEVICE fp32 tensor (the reference's `final_alpha_cumprod.to(device)`) with fp16 /
fp32 tensors?  Each op of the `t - skip < 0` DDIM step is evaluated by torch on the GPU and compared with candidate rules
evaluated on the CPU from the same bits; prints which rule reproduces it."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cfgpp_amd.schedule import SchedulerTables
tb = SchedulerTables(50)
a_cpu = tb.final_alpha_cumprod.clone()
a_dev = a_cpu.cuda()
print("final alpha", float(a_cpu), "sqrt cpu", (a_cpu.sqrt()).view(torch.int32).item(), "sqrt dev", a_dev.sqrt().cpu().view(torch.int32).item(),
      "sqrt(1-a) cpu", (1 - a_cpu).sqrt().view(torch.int32).item(), "dev", (1 - a_dev).sqrt().cpu().view(torch.int32).item(),
      "pinned", torch.tensor(tb.ddim_sqrt_coeffs(1)[2]).view(torch.int32).item(), torch.tensor(tb.ddim_sqrt_coeffs(1)[3]).view(torch.int32).item())
g = torch.Generator().manual_seed(0)
xh = torch.randn(1 << 16, generator=g).half(); xf = torch.randn(1 << 16, generator=g)
F, H = torch.float32, torch.float16
s_dev = (1 - a_dev).sqrt(); s = s_dev.cpu()            # use the DEVICE's own sqrt value in every candidate
r_dev = a_dev.sqrt(); r = r_dev.cpu()


def report(name, got, cands):
    got = got.cpu()
    hits = [k for k, v in cands.items() if v.dtype == got.dtype and torch.equal(v, got)]
    print(f"{name:34s} dtype {str(got.dtype):14s} matches: {hits or 'NONE'}   " + ", ".join(f"{k}: {int((v.to(got.dtype) != got).sum())}" for k, v in cands.items()))


with torch.autocast(device_type="cuda", dtype=torch.float16):
    report("s_dev * x_half", s_dev * xh.cuda(), {"fp32 scalar": (xh.float() * s).half(), "fp16-rounded scalar": (xh.float() * s.half().float()).half()})
    report("x_half * s_dev", xh.cuda() * s_dev, {"fp32 scalar": (xh.float() * s).half(), "fp16-rounded scalar": (xh.float() * s.half().float()).half()})
    report("r_dev * x_float", r_dev * xf.cuda(), {"fp32": xf * r})
    report("r_dev * x_half", r_dev * xh.cuda(), {"fp32 scalar": (xh.float() * r).half(), "fp16-rounded scalar": (xh.float() * r.half().float()).half()})
    report("x_float / r_dev", xf.cuda() / r_dev, {"true division": xf / r, "reciprocal": xf * (torch.tensor(1.0) / r)})
    report("x_half / r_dev", xh.cuda() / r_dev, {"true div fp32 divisor": (xh.float() / r).half(), "true div fp16 divisor": (xh.float() / r.half().float()).half(),
                                                  "reciprocal fp32": (xh.float() * (torch.tensor(1.0) / r)).half(),
                                                  "reciprocal of fp16 divisor": (xh.float() * (torch.tensor(1.0) / r.half().float())).half()})
    report("x_float - s_dev * x_half", xf.cuda() - s_dev * xh.cuda(), {"fp32 scalar": xf - (xh.float() * s).half().float(), "fp16 scalar": xf - (xh.float() * s.half().float()).half().float()})
