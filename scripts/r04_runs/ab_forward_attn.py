"""Whole UNet forwards with the self-attention kernel switched IN ONE ENGINE (same tiles, same buffers, same process):
    python scripts/r04_runs/ab_forward_attn.py sd15 8 [modes, default 1,2,3]
mode 1 = flash loop everywhere, 2 = woven kernel for >= 2048 keys, 3 = woven kernel for >= 256 keys.  Rounds of 20 back-to-back
predict() calls per mode, the modes interleaved, 6 rounds; prints min / median per mode and the difference of the outputs."""
import os, statistics, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfgpp_amd import _lib
from cfgpp_amd.hip_engine import HipEngine
from cfgpp_amd.unet_config import CONFIGS
from cfgpp_amd.weights import synth_state_dict
name, B = sys.argv[1], int(sys.argv[2])
modes = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
cfg = CONFIGS[name]; lib = _lib.load()
cache = f"/tmp/cfgpp_synth_{name}.safetensors"
if not os.path.exists(cache):
    from safetensors.torch import save_file
    save_file({k: v.half().contiguous() for k, v in synth_state_dict(cfg, 0).items()}, cache + ".tmp"); os.replace(cache + ".tmp", cache)
eng = HipEngine(name, max_batch=B, weights=cache)
g = torch.Generator().manual_seed(0)
uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
te = ti = None
if cfg.addition_embed:
    te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
    ti = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B)).cuda()
eng.set_context(uc, c, te, ti)
z = torch.randn(B, 4, eng.H, eng.W, generator=g).cuda()
for _ in range(3): eng.predict(z, 500.0)          # tunes
torch.cuda.synchronize()
outs, times = {}, {m: [] for m in modes}
for m in modes:
    lib.cfgpp_attention_set_dma(m)
    eu, ec = eng.predict(z, 500.0); outs[m] = torch.cat([eu, ec]).float().clone()
for rnd in range(6):
    for m in modes:
        lib.cfgpp_attention_set_dma(m)
        eng.predict(z, 500.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): eng.predict(z, 500.0)
        e1.record(); torch.cuda.synchronize()
        times[m].append(e0.elapsed_time(e1) / 20)
fam = {}
for m in modes:                      # per-family sums of a profiled forward (every launch timed on its own), median of 3
    lib.cfgpp_attention_set_dma(m)
    runs = [eng.unet.profile(z, 500.0) for _ in range(3)]
    fam[m] = {k: statistics.median(r[k]["ms"] for r in runs) for k in runs[0] if k != "detail"}
lib.cfgpp_attention_set_dma(1)
base = outs[modes[0]]
for m in modes:
    print(f"{name} B={B} attention mode {m}: forward wall min {min(times[m]):.3f} ms median {statistics.median(times[m]):.3f} ms  rounds {[round(t, 3) for t in times[m]]}"
          f"  rel-L2 vs mode {modes[0]}: {float((outs[m] - base).norm() / base.norm()):.2e}   families ms (profiled): " + ", ".join(f"{k} {v:.3f}" for k, v in fam[m].items()), flush=True)
