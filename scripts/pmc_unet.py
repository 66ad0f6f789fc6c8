"""UNet-only forwards at the benchmark batch - the population bench.py's `roofline.achieved` is computed on - as a
rocprofv3 target.  Un-profiled:  python scripts/pmc_unet.py sd15 16 --save-hints   (tunes, writes the per-launch tiles)
Profiled:  rocprofv3 --pmc FETCH_SIZE -- python scripts/pmc_unet.py sd15 16 --load-hints [--iters 5]"""
import argparse, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("cfg"); ap.add_argument("rows", type=int)
ap.add_argument("--save-hints", action="store_true"); ap.add_argument("--load-hints", action="store_true"); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
from cfgpp_amd import _lib
from cfgpp_amd.hip_engine import HipEngine
hf = os.path.join(ROOT, "gpurun_out", f"tune_{a.cfg}_rows{a.rows}.json")
if a.load_hints: _lib.load().cfgpp_igemm_set_autotune(1)
# the synthetic state dict is generated once per box and cached (SDXL: ~45 s of CPU RNG per process otherwise; five processes per config)
from cfgpp_amd.unet_config import CONFIGS
from cfgpp_amd.weights import synth_state_dict
cache = f"/tmp/cfgpp_synth_{a.cfg}.safetensors"
if not os.path.exists(cache):
    from safetensors.torch import save_file
    save_file({k: v.half().contiguous() for k, v in synth_state_dict(CONFIGS[a.cfg], 0).items()}, cache + ".tmp"); os.replace(cache + ".tmp", cache)
eng = HipEngine(a.cfg, max_batch=a.rows // 2, weights=cache)
cfg = eng.cfg; B = a.rows // 2
g = torch.Generator().manual_seed(0)
uc = torch.randn(1, 77, cfg.cross_attention_dim, generator=g).half() * 0.5; c = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).half() * 0.5
te = ti = None
if cfg.addition_embed:
    te = torch.randn(a.rows, cfg.addition_pooled_dim, generator=g).half() * 0.5; ti = torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * a.rows)
eng.set_context(uc.cuda(), c.cuda(), te, ti)
z = torch.randn(B, 4, eng.H, eng.W, generator=g).cuda()
if a.load_hints:
    eng.unet.import_tuning(json.load(open(hf)), a.rows)
for _ in range(a.iters): eng.predict(z, 500.0)
torch.cuda.synchronize()
if a.save_hints:
    os.makedirs(os.path.dirname(hf), exist_ok=True); json.dump(eng.unet.export_tuning(a.rows), open(hf, "w")); print("wrote", hf)
    pr = eng.unet.profile(z, 500.0, detail=True)
    open(os.path.join(ROOT, "gpurun_out", f"detail_{a.cfg}_rows{a.rows}.txt"), "w").write(pr["detail"])
