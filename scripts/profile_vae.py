"""Per-launch profile of one VAE decode, aggregated by (family, shape): python scripts/profile_vae.py [batch] [latent_hw]
(batch 8 @ 64 = the SD1.5 bench job's decode, batch 8 @ 128 = the Lightning job's, batch 2 @ 128 = SDXL's)"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cfgpp_amd.vae import HipVAE
from cfgpp_amd import _lib
if os.environ.get("CFGPP_GN_PRESTATS"): _lib.load().cfgpp_groupnorm_set_prestats(int(os.environ["CFGPP_GN_PRESTATS"]))
if os.environ.get("TUNE_MASK"): _lib.load().cfgpp_igemm_set_tune_mask(int(os.environ["TUNE_MASK"], 0))      # e.g. 0xffffffff: also offer the big4 tiles
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
vae = HipVAE(0.18215, (hw, hw), max_batch=B, with_encoder=False)
z = torch.randn(B, 4, hw, hw, device="cuda") * 0.18215
for _ in range(2): vae.decode_image(z)
agg = collections.OrderedDict(); N = 3
for _ in range(N):
    for kind, desc, us, gf in vae.profile(z):
        a = agg.setdefault((kind, desc), [0, 0.0, 0.0]); a[0] += 1; a[1] += us; a[2] += gf
tot = sum(a[1] for a in agg.values()) / N
print(f"# vae decode batch={B} latent={hw}x{hw} total {tot/1e3:.2f} ms  ({sum(a[2] for a in agg.values())/N/tot*1e3:.0f} TF/s over the whole decode)")
for (kind, desc), (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us/N/1e3:8.3f} ms  {100*us/N/tot:5.1f}%  x{cnt//N:<3d} {gf/us*1e3 if us else 0:7.1f} TF/s  [{kind}] {desc}")
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): vae.decode_image(z)
e1.record(); torch.cuda.synchronize()
print(f"# wall {e0.elapsed_time(e1)/5:.2f} ms/decode (5 back-to-back decode_image() calls)")
