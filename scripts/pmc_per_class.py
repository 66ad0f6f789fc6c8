"""HBM traffic per implicit-GEMM LAUNCH CLASS of a UNet forward: which launches over-fetch?
    python scripts/pmc_per_class.py --fetch DIR --write DIR --detail gpurun_out/detail_<cfg>_rows<R>.txt --rows R
DIR = rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE output over scripts/pmc_unet.py <cfg> <R> --load-hints (UNet-only forwards with the
pinned tiles).  The igemm-family dispatches of a process are, in order: the set_context projections (cross-attention K / V^T, once),
then F identical forwards of L launches each; dispatch i of a forward is launch i of the plan's igemm lines in the detail file, so
counters are folded onto (kind, shape) classes by position.  K-split launches add one igemm_reduce dispatch, which is its own kernel
family and not counted here.  FETCH_SIZE is doubled (gfx950: 128-B requests tallied as 64 B, MI355X_MICROARCH.md), both are KiB."""
import argparse, collections, csv, glob, os, re

ap = argparse.ArgumentParser()
for k in ("fetch", "write", "detail"): ap.add_argument("--" + k, required=True)
ap.add_argument("--rows", type=int, required=True)
a = ap.parse_args()


def is_igemm(k):
    return ("igemm_kernel" in k or "igemm16_kernel" in k or "tile32_kernel" in k or "big4_kernel" in k or "big4p_kernel" in k) and "igemm_reduce" not in k


def per_dispatch(d, counter):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and is_igemm(r["Kernel_Name"]):
                rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


launches = []
for line in open(a.detail):
    p = line.rstrip("\n").split("\t")
    if len(p) >= 4 and p[1] == "0":
        launches.append(p[2])
L = len(launches)


def alg_bytes(d):
    m = re.search(r"HW=(\d+) N=(\d+) K=(\d+)", d)
    HW, N, K = (int(x) for x in m.groups()); M = a.rows * HW
    if d.startswith("conv3x3"):
        am = int(re.search(r"amode=(\d)", d).group(1)); cin = K // 9
        a_b = M * cin * 2 * (4 if am == 2 else 0.25 if am == 3 else 1)
    else:
        a_b = M * K * 2
    out_b = M * (N // 2 if d.startswith("geglu") else N) * 2
    return a_b + N * K * 2 + out_b + (M * N * 2 if "+res" in d else 0)


fe, wr = per_dispatch(a.fetch, "FETCH_SIZE"), per_dispatch(a.write, "WRITE_SIZE")
agg = collections.OrderedDict()
for name, vals, scale in (("fetch", fe, 2048.0), ("write", wr, 1024.0)):
    ctx = len(vals) % L
    body = vals[ctx:]
    F = len(body) // L
    for i, v in enumerate(body[:F * L]):
        c = agg.setdefault(launches[i % L], {"n": 0, "fetch": 0.0, "write": 0.0, "nf": 0, "nw": 0})
        c[name] += v * scale
        c["nf" if name == "fetch" else "nw"] += 1
    print(f"# {name}: {len(vals)} igemm dispatches = {ctx} at set_context + {F} forwards x {L} launches")
tot_m = tot_a = 0.0
rows = []
for d, c in agg.items():
    per = c["fetch"] / max(c["nf"], 1) + c["write"] / max(c["nw"], 1)
    cnt = launches.count(d)
    rows.append((per * cnt, cnt, per, alg_bytes(d), d))
    tot_m += per * cnt; tot_a += alg_bytes(d) * cnt
print(f"# per forward: measured {tot_m / 1e6:.0f} MB vs algorithmic {tot_a / 1e6:.0f} MB = {tot_m / tot_a:.2f}x; classes by measured bytes per forward")
print(f"{'MB/forward':>11s} {'x':>4s} {'MB/launch':>10s} {'algorithmic':>12s} {'ratio':>6s}  launch class")
for t, cnt, per, alg, d in sorted(rows, reverse=True):
    print(f"{t / 1e6:11.1f} {cnt:4d} {per / 1e6:10.2f} {alg / 1e6:12.2f} {per / alg:6.2f}  {d}")
