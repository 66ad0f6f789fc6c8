"""Same-box, same-process A/B of whole UNet forwards: python scripts/r04_runs/ab_forward.py sd15 8 "0:1,2:1,0:2,2:2,0:4"
variants = fuse_ln:lanes.  The synthetic state dict is generated once; every variant builds its own engine from it, tunes,
and is timed as back-to-back predict() calls (3 x 20 forwards: min and median) - the number the sampling loop sees.  For
single-lane variants the per-family sums of a profiled forward are printed as well."""
import gc
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.hip_engine import HipEngine  # noqa: E402
from cfgpp_amd.unet_config import CONFIGS  # noqa: E402
from cfgpp_amd.weights import synth_state_dict  # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2])
variants = [tuple(int(x) for x in v.split(":")) for v in sys.argv[3].split(",")]
cfg = CONFIGS[name]
lib = _lib.load()
t0 = time.time()
sd = {k: v.half() for k, v in synth_state_dict(cfg, 0).items()}
print(f"# {name} B={B} (UNet rows {2 * B}); state dict in {time.time() - t0:.1f} s", flush=True)
g = torch.Generator().manual_seed(0)
uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
te = ti = None
if cfg.addition_embed:
    te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
    ti = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B)).cuda()
ref = None
for fuse, lanes in variants:
    lib.cfgpp_unet_set_fuse_ln(fuse)
    t0 = time.time()
    eng = HipEngine(cfg, max_batch=B, weights=sd, lanes=lanes)
    eng.set_context(uc, c, te, ti)
    z = torch.randn(B, 4, eng.H, eng.W, generator=g.manual_seed(1)).cuda()
    for _ in range(3):
        eu, ec = eng.predict(z, 500.0)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    out = torch.cat([eu, ec]).float()
    if ref is None:
        ref = out.clone()
    rel = float((out - ref).norm() / ref.norm())
    times = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.predict(z, 500.0)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 20)
    line = f"fuse_ln={fuse} lanes={eng.lanes}: forward wall min {min(times):.3f} ms  median {statistics.median(times):.3f} ms   rel-L2 vs first variant {rel:.2e}   (engine + tuning {build_s:.0f} s)"
    if eng.lanes == 1:
        pr = eng.unet.profile(z, 500.0)
        line += "   families ms: " + ", ".join(f"{k} {v['ms']:.2f} ({v['launches']})" for k, v in pr.items())
    print(line, flush=True)
    del eng, eu, ec
    gc.collect()
    torch.cuda.empty_cache()
