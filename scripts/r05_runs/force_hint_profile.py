"""In-situ per-launch times of ONE forced tile config against the tuned plan, same engine, same box:
    python scripts/r05_runs/force_hint_profile.py sd15 8 25 [24 26 ...]
Every igemm launch of the plan gets cfg_hint = H (launches that do not admit H run their heuristic tile; rule-based launches -
K-split, the 16x16x32 tile - ignore hints unless MF16=0 / ROUNDS=1 hands them to the hint path); three profiled forwards per
variant, per-launch minimum.  Prints the launches where H is faster / slower than the tuned plan."""
import collections
import os
import sys

os.environ["CFGPP_TUNE_CACHE"] = "0"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.hip_engine import HipEngine  # noqa: E402
from cfgpp_amd.unet_config import CONFIGS  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
forced = [int(x) for x in sys.argv[3:]]
cfg = CONFIGS[name]
lib = _lib.load()
lib.cfgpp_igemm_set_mf16(int(os.environ.get("MF16", "4")))
lib.cfgpp_igemm_set_mf16_rounds(int(os.environ.get("ROUNDS", "2")))
eng = HipEngine(cfg, max_batch=B)
g = torch.Generator().manual_seed(0)
uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
te = ti = None
if cfg.addition_embed:
    te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
    ti = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B)).cuda()
eng.set_context(uc, c, te, ti)
z = torch.randn(B, 4, eng.H, eng.W, generator=g).cuda()
for _ in range(3):
    eng.predict(z, 500.0)
torch.cuda.synchronize()
tuned = eng.export_tuning()


def table():
    agg = collections.OrderedDict()
    for _ in range(3):
        for line in eng.unet.profile(z, 500.0, detail=True)["detail"].strip().split("\n"):
            i, kind, desc, us, gf = line.split("\t")
            a = agg.setdefault(int(i), [kind, desc, 1e30, float(gf)])
            a[2] = min(a[2], float(us))
    return agg


def wall():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.predict(z, 500.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


base = table()
print(f"# {name} B={B}: tuned plan forward {wall():.3f} ms; igemm sum {sum(v[2] for v in base.values() if v[0] == '0') / 1e3:.3f} ms; pins " +
      ", ".join(f"{k}:{v}" for k, v in sorted(collections.Counter(h & 63 for h in tuned).items())), flush=True)
for H in forced:
    eng.import_tuning([H] * len(tuned), B)
    eng.predict(z, 500.0)
    torch.cuda.synchronize()
    t = table()
    print(f"## every launch hinted {H}: forward {wall():.3f} ms; igemm sum {sum(v[2] for v in t.values() if v[0] == '0') / 1e3:.3f} ms")
    rows = collections.OrderedDict()
    for i, (kind, desc, us, gf) in t.items():
        if kind != "0":
            continue
        r = rows.setdefault(desc, [0, 0.0, 0.0, gf])
        r[0] += 1; r[1] += us; r[2] += base[i][2]
    for desc, (n, us, us0, gf) in sorted(rows.items(), key=lambda kv: kv[1][1] - kv[1][2]):
        if abs(us - us0) > 0.02 * us0:
            print(f"   {us / n:8.1f} us vs tuned {us0 / n:8.1f} us  ({gf / (us / n) * 1e3:7.1f} vs {gf / (us0 / n) * 1e3:7.1f} TF/s)  x{n:<3d} {desc}")
eng.import_tuning(tuned, B)
