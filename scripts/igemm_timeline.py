"""Where the time of ONE implicit-GEMM launch goes, inside a real forward: python scripts/igemm_timeline.py sdxl 4 "linear HW=1024 N=1280 K=1280 +res" ...

For every description given (a (kind, shape) line of scripts/profile_unet.py; the first launch of the plan that matches) the
launch is re-run in situ with the kernel's timeline armed (cfgpp_igemm_timeline): every workgroup stamps s_memtime at entry,
when its first K-tile has landed, after the first K-tile, after the K loop and after its stores.  Printed per launch: clock,
kernel span, start skew, and the distribution over workgroups of prologue / per-K-tile / epilogue time."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.hip_engine import HipEngine  # noqa: E402

name, rows = sys.argv[1], int(sys.argv[2])
wanted = sys.argv[3:]
lib = _lib.load()
lib.cfgpp_igemm_set_mf16_heads(int(os.environ.get("MF16_HEADS", "1")))
lib.cfgpp_igemm_set_mf16_rounds(int(os.environ.get("MF16_ROUNDS", "2")))
eng = HipEngine(name, max_batch=rows // 2)
cfg, B = eng.cfg, rows // 2
uc = torch.randn(1, 77, cfg.cross_attention_dim).half() * 0.5; c = torch.randn(B, 77, cfg.cross_attention_dim).half() * 0.5
te = ti = None
if cfg.addition_embed:
    te = torch.randn(rows, cfg.addition_pooled_dim).half() * 0.5; ti = torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * rows)
eng.set_context(uc.cuda(), c.cuda(), te, ti)
z = torch.randn(B, 4, eng.H, eng.W, device="cuda")
for _ in range(3): eng.predict(z, 500.0)
if os.environ.get("FORCE_HINT"):      # every launch hinted with ONE tile config (launches that do not admit it keep their heuristic tile)
    eng.import_tuning([int(os.environ["FORCE_HINT"])] * len(eng.export_tuning()), B)
    for _ in range(2): eng.predict(z, 500.0)
detail = [ln.split("\t") for ln in eng.unet.profile(z, 500.0, detail=True)["detail"].strip().split("\n")]
ordinal, k = {}, 0
times = {}
for i, kind, desc, us, gf in detail:
    if kind == "0":
        ordinal.setdefault(desc, k); k += 1
        times.setdefault(desc, []).append(float(us))
CAP = 8192
buf = torch.zeros((CAP, 16), dtype=torch.int64, device="cuda")
info = (C.c_int * 12)()


def pct(a, q):
    return float(np.percentile(a, q))


for want in wanted:
    hits = [d for d in ordinal if want in d]
    if not hits:
        print(f"## no launch matches {want!r}"); continue
    desc = hits[0]
    buf.zero_()
    torch.cuda.synchronize()
    lib.cfgpp_igemm_timeline(buf.data_ptr(), CAP, ordinal[desc])
    eng.predict(z, 500.0)
    torch.cuda.synchronize()
    lib.cfgpp_igemm_timeline(None, 0, -1)
    lib.cfgpp_igemm_timeline_info(info)
    tile, grid, thr, BM, BN, NST, ksplit, nmaj, M, N, K, epi = list(info)
    r = buf[:grid].cpu().numpy().astype(np.float64)
    ok = r[:, 3] > 0
    if not ok.any():
        print(f"## {desc}: nothing recorded (grid {grid})"); continue
    r = r[ok]
    t0, t1, t2, t3, rt0, rt1, t7, t8, t9 = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 7], r[:, 8], r[:, 9]
    ghz = float(np.median((t3 - t0) / np.maximum(rt1 - rt0, 1))) * 0.1          # memtime ticks per 10 ns
    span_us = (rt1.max() - rt0.min()) * 0.01
    nk = (K // 64) // max(ksplit, 1)
    cyc = lambda a: a                                                            # noqa: E731  (ticks = shader cycles)
    pro, first, loop, epi_c, tot = cyc(t1 - t0), cyc(t7 - t1), cyc(t2 - t1), cyc(t3 - t2), cyc(t3 - t0)
    per_tile = (t2 - t7) / max(nk - 1, 1)
    flops = 2.0 * M * N * K
    xcc = (buf[:grid, 6].cpu().numpy()[ok] >> 32) & 0xF
    print(f"## {desc}   event time {np.mean(times[desc]):.1f} us x{len(times[desc])}")
    print(f"   tile id {tile} ({BM}x{BN}, {thr // 64} waves, {NST} stages) grid {grid} ksplit {ksplit} n_major {nmaj}  M {M} N {N} K {K} ({nk} K-tiles/WG) epi {epi}")
    print(f"   clock ~{ghz:.2f} GHz; span first entry -> last exit {span_us:.1f} us = {flops / span_us / 1e6:.0f} TF/s; MFMA floor/K-tile "
          f"{BM * BN * 64 * 2 / (2.5e15 / 256) * 2.4e9:.0f} cyc @2.4GHz-peak")
    print(f"   start skew (us after first entry): p50 {pct((rt0 - rt0.min()) * 0.01, 50):.2f} p90 {pct((rt0 - rt0.min()) * 0.01, 90):.2f} max {((rt0 - rt0.min()) * 0.01).max():.2f}")
    for nm, a in (("  entry -> first DMA issue (kernargs, index math)", t8 - t0), ("  issuing the prologue's DMAs", t9 - t8), ("  issued -> first tile landed", t1 - t9),
                  ("prologue (entry -> first tile landed)", pro), ("first K-tile", first), ("per K-tile (steady)", per_tile), ("K loop total", loop),
                  ("epilogue (-> stores done)", epi_c), ("workgroup total", tot)):
        print(f"   {nm:50s} cycles p10 {pct(a, 10):9.0f}  p50 {pct(a, 50):9.0f}  p90 {pct(a, 90):9.0f}  max {a.max():9.0f}   (p50 = {pct(a, 50) / ghz / 1e3:.2f} us)")
    print(f"   workgroups per XCC id: {np.bincount(xcc.astype(np.int64), minlength=8).tolist()}")
    # workgroup turnover per CU: HW_ID bits 8..15 = (cu, sh, se) inside the XCC; s_memrealtime ticks are 10 ns
    hw = buf[:grid, 6].cpu().numpy()[ok] & 0xFFFFFFFF
    cu = (xcc.astype(np.int64) << 8) | ((hw.astype(np.int64) >> 8) & 0xFF)
    gaps, per_cu = [], []
    for key in np.unique(cu):
        sel = np.where(cu == key)[0]
        order = sel[np.argsort(rt0[sel])]
        per_cu.append(len(order))
        for x, y in zip(order[:-1], order[1:]):
            gaps.append((rt0[y] - rt1[x]) * 0.01)
    busy = float(((rt1 - rt0) * 0.01).sum()) / (len(per_cu) * max(span_us, 1e-9))
    print(f"   CUs seen {len(per_cu)} (workgroups per CU min {min(per_cu)} max {max(per_cu)}); resident-workgroup time / (CUs x span) = {busy:.2f}")
    if gaps:
        g = np.array(gaps)
        print(f"   exit -> next entry on the same CU (us): p10 {pct(g, 10):.2f} p50 {pct(g, 50):.2f} p90 {pct(g, 90):.2f}  (negative = two workgroups resident)")
