"""Same-box, same-process A/B of whole UNet forwards:
    python scripts/r06_runs/ab_forward.py sd15 8 "base:mask=0xffff7fff;lin32:mask=0xffffffff;lin32_nomf16lin:mask=0xffffffff,mf16lin=0"
variant = name:key=value,...  keys: mask (tuner candidate mask, bit c = tile config c), mf16lin (16x16x32 rule takes linears),
rounds (that rule also takes grids of 2 .. n full rounds), mf16 (0 = rule off, 3 / 4 = stages), prestats (GroupNorm takes the
producers' statistics: 1 / 0).
The synthetic state dict is generated once; every variant builds its own engine from it, tunes, and is timed as back-to-back
predict() calls (3 x 20 forwards: min and median) - the number the sampling loop sees - plus the per-family sums and the pinned
tile histogram of a profiled forward; `--table` prints the per-launch table of the LAST variant."""
import collections
import gc
import os
import statistics
import sys
import time

os.environ["CFGPP_TUNE_CACHE"] = "0"      # every variant tunes for itself: no pins from disk (cfgpp_amd/tune_cache.py)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.hip_engine import HipEngine  # noqa: E402
from cfgpp_amd.unet_config import CONFIGS  # noqa: E402
from cfgpp_amd.weights import synth_state_dict  # noqa: E402

name = sys.argv[1]
B = int(sys.argv[2])
variants = []
for v in sys.argv[3].split(";"):
    vn, _, kv = v.partition(":")
    variants.append((vn, dict(x.split("=") for x in kv.split(",") if x)))
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
for vn, kv in variants:
    lib.cfgpp_igemm_set_tune_mask(int(kv.get("mask", "0xffffffff"), 0))
    lib.cfgpp_igemm_set_mf16_linear(int(kv.get("mf16lin", "1")))
    lib.cfgpp_igemm_set_mf16_rounds(int(kv.get("rounds", "2")))
    lib.cfgpp_igemm_set_mf16(int(kv.get("mf16", "4")))
    lib.cfgpp_groupnorm_set_prestats(int(kv.get("prestats", "1")))
    lib.cfgpp_igemm_set_blocked_walk(int(kv.get("blocked", "1")))
    t0 = time.time()
    eng = HipEngine(cfg, max_batch=B, weights=sd)
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
    pr = eng.unet.profile(z, 500.0, detail=True)
    pins = collections.Counter(h & 63 for h in eng.unet.export_tuning())
    print(f"{vn} {kv}: forward wall min {min(times):.3f} ms  median {statistics.median(times):.3f} ms   rel-L2 vs first variant {rel:.2e}   "
          f"(engine + tuning {build_s:.0f} s)   families ms: " + ", ".join(f"{k} {v['ms']:.2f} ({v['launches']})" for k, v in pr.items() if k != "detail")
          + "   pins: " + ", ".join(f"{k}:{v}" for k, v in sorted(pins.items())), flush=True)
    if "--table" in sys.argv and (vn, kv) == variants[-1]:
        agg = collections.OrderedDict()
        for _ in range(3):
            for line in eng.unet.profile(z, 500.0, detail=True)["detail"].strip().split("\n"):
                i, kind, desc, us, gf = line.split("\t")
                a = agg.setdefault((kind, desc), [0, 0.0, 0.0]); a[0] += 1; a[1] += float(us); a[2] += float(gf)
        tot = sum(a[1] for a in agg.values()) / 3
        print(f"# per-launch table of variant {vn}: total {tot / 1e3:.2f} ms/forward")
        for (kind, desc), (cnt, us, gf) in sorted(agg.items(), key=lambda kv_: -kv_[1][1]):
            print(f"{us / 3 / 1e3:8.3f} ms  {100 * us / 3 / tot:5.1f}%  x{cnt // 3:<3d} {gf / us * 1e3 if us else 0:7.1f} TF/s  [{kind}] {desc}")
    del eng, eu, ec
    gc.collect()
    torch.cuda.empty_cache()
