"""(NOT SHIPPED: the two-wave variant lost this A/B and was removed again - the script needs commit 84add5e's attn_kernel.hip.)  Same engine, same box: dp = 64 flash attention with 128-query workgroups only (attn_nw = 4, rounds 2-5) against the grid rule
that switches unevenly loaded grids to 64-query two-wave workgroups (attn_nw = 0, default), interleaved rounds of 20 forwards.
    python scripts/r06_runs/ab_attn_nw.py sdxl 2 [rounds]"""
import os
import statistics
import sys

os.environ["CFGPP_TUNE_CACHE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.hip_engine import HipEngine  # noqa: E402
from cfgpp_amd.unet_config import CONFIGS  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = CONFIGS[name]
lib = _lib.load()
g = torch.Generator().manual_seed(0)
uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
te = ti = None
if cfg.addition_embed:
    te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
    ti = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B)).cuda()
eng = HipEngine(cfg, max_batch=B)
eng.set_context(uc, c, te, ti)
z = torch.randn(B, 4, eng.H, eng.W, generator=g).cuda()
outs = {}
for nw in (4, 0):
    lib.cfgpp_attention_set_waves(nw)
    for _ in range(3):
        eu, ec = eng.predict(z, 500.0)
    torch.cuda.synchronize()
    outs[nw] = torch.cat([eu, ec]).clone()
print(f"# {name} B={B} rows {2 * B} build {_lib.build_id()}; eps bit-identical across workgroup shapes: {bool(torch.equal(outs[4], outs[0]))}", flush=True)
res = {4: [], 0: []}
for r in range(rounds):
    for nw in (4, 0):
        lib.cfgpp_attention_set_waves(nw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.predict(z, 500.0)
        e1.record()
        torch.cuda.synchronize()
        res[nw].append(e0.elapsed_time(e1) / 20)
for nw, label in ((4, "four-wave only"), (0, "grid rule    ")):
    lib.cfgpp_attention_set_waves(nw)
    pr = eng.unet.profile(z, 500.0)
    print(f"{label}: forward ms min {min(res[nw]):.3f} median {statistics.median(res[nw]):.3f}  all " + " ".join(f"{x:.3f}" for x in res[nw]) +
          f"   attention family {pr['attention']['ms']:.3f} ms ({pr['attention']['launches']} launches, {pr['attention']['flops'] / pr['attention']['ms'] / 1e9:.0f} TF/s)", flush=True)
lib.cfgpp_attention_set_waves(0)
print(f"grid rule vs four-wave only: {(min(res[0]) - min(res[4])) / min(res[4]) * 100:+.2f} % per forward")
