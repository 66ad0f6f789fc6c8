"""Per-launch profile of one UNet forward, aggregated by (family, shape): python scripts/profile_unet.py sd15 16"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cfgpp_amd.hip_engine import HipEngine
from cfgpp_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "sd15"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hw = int(sys.argv[3]) if len(sys.argv) > 3 else None
staging = int(os.environ.get("STAGING", "1"))
_lib.load().cfgpp_igemm_set_staging(staging)
_lib.load().cfgpp_igemm_set_autotune(int(os.environ.get("AUTOTUNE", "1")))
# A/B switches: TUNE_MASK=0x5f2 = the round-2 mid-round candidate set without the tile-walk stage; BIG_SPLIT=0 turns the
# big-tile K-split rule off; LN_RPW = LayerNorm rows per wave
if os.environ.get("TUNE_MASK"):      # default: the library's own candidate set (cfgpp_igemm_set_tune_mask)
    _lib.load().cfgpp_igemm_set_tune_mask(int(os.environ["TUNE_MASK"], 0))
_lib.load().cfgpp_igemm_set_big_split(int(os.environ.get("BIG_SPLIT", "0")))
_lib.load().cfgpp_layernorm_set_rows_per_wave(int(os.environ.get("LN_RPW", "0")))
_lib.load().cfgpp_igemm_set_mf16_heads(int(os.environ.get("MF16_HEADS", "1")))       # 1: head-major epilogue of that tile (unvalidated)
_lib.load().cfgpp_igemm_set_mf16_rounds(int(os.environ.get("MF16_ROUNDS", "2")))     # > 1: also 2 .. n full rounds of 256 tiles
_lib.load().cfgpp_igemm_set_mf16(int(os.environ.get("MF16", "4")))                 # 3 / 4: 16x16x32-MFMA 128x160 tile by rule
_lib.load().cfgpp_igemm_set_split_tile(int(os.environ.get("SPLIT_TILE", "14")))
_lib.load().cfgpp_attention_set_stagger(int(os.environ.get("ATTN_STAGGER", "0")))
_lib.load().cfgpp_attention_set_cross(int(os.environ.get("ATTN_CROSS", "1")))
_lib.load().cfgpp_attention_set_dma(int(os.environ.get("ATTN_MODE", "1")))
_lib.load().cfgpp_igemm_set_tail_split(int(os.environ.get("TAIL_SPLIT", "1")))      # 2 = round-1 slice count (rounded up)
eng = HipEngine(name, max_batch=rows // 2, latent_hw=(hw, hw) if hw else None)
cfg = eng.cfg
B = rows // 2
uc = torch.randn(1, 77, cfg.cross_attention_dim).half() * 0.5; c = torch.randn(B, 77, cfg.cross_attention_dim).half() * 0.5
te = ti = None
if cfg.addition_embed:
    te = torch.randn(rows, cfg.addition_pooled_dim).half() * 0.5; ti = torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * rows)
eng.set_context(uc.cuda(), c.cuda(), te, ti)
z = torch.randn(B, 4, eng.H, eng.W, device="cuda")
for _ in range(2): eng.predict(z, 500.0)
agg = collections.OrderedDict()
N = 3
for _ in range(N):
    pr = eng.unet.profile(z, 500.0, detail=True)
    for line in pr["detail"].strip().split("\n"):
        i, kind, desc, us, gf = line.split("\t")
        a = agg.setdefault((kind, desc), [0, 0.0, 0.0]); a[0] += 1; a[1] += float(us); a[2] += float(gf)
tot = sum(a[1] for a in agg.values()) / N
print(f"# {name} rows={rows} staging={'glds' if staging else 'reg'} total {tot/1e3:.2f} ms/forward")
for (kind, desc), (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us/N/1e3:8.3f} ms  {100*us/N/tot:5.1f}%  x{cnt//N:<3d} {gf/us*1e3 if us else 0:7.1f} TF/s  [{kind}] {desc}")
# wall clock of back-to-back forwards (launch gaps included) vs the sum of per-launch times above
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): eng.predict(z, 500.0)
e1.record(); torch.cuda.synchronize()
print(f"# wall {e0.elapsed_time(e1)/10:.2f} ms/forward (10 back-to-back predict() calls)")
pins = collections.Counter(eng.unet.export_tuning())
print("# pinned hints (tile config | walk << 6 : launches): " + ", ".join(f"{k}:{v}" for k, v in sorted(pins.items())))
