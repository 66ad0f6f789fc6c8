"""Combine the rocprofv3 passes over scripts/pmc_unet.py (UNet-only forwards at the benchmark batch) into the per-family
summary bench.py reads:  python scripts/pmc_summary.py --fetch DIR --write DIR --sq DIR [--trace DIR] --detail FILE --rows R --out JSON
FETCH_SIZE / WRITE_SIZE are KiB (rocprofv3); on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads as
64 B, so it is DOUBLED (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as is.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES /
(4 SIMDs x SQ_BUSY_CU_CYCLES).  Durations: rocprofv3 --kernel-trace --stats of the same command (--trace DIR)."""
import argparse, collections, csv, glob, json, os, re, sys

def family(k):
    if "igemm_reduce" in k: return "igemm_reduce"
    if "igemm_kernel" in k or "igemm16_kernel" in k or "tile32_kernel" in k or "big4_kernel" in k or "big4p_kernel" in k: return "igemm"   # every kernel of the implicit-GEMM family
    if "attn64_kernel" in k or "xattn64_kernel" in k or "attn_kernel" in k: return "attention"
    if "gn_" in k: return "groupnorm"
    if "layernorm" in k: return "layernorm"
    if "ddim_step" in k or "kdiff" in k or "lincomb" in k: return "step"
    return "other"

def load(d):
    """family -> counter -> [n, sum], family -> [n, sum_ns]"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            a = agg[fam][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); dur[fam][0] += 1; dur[fam][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dur

def algorithmic_bytes(detail_file, rows):
    """unique A + W + out (+ residual) bytes of every igemm launch of one forward (fp16), summed, and the launch count"""
    tot, n = 0.0, 0
    for line in open(detail_file):
        p = line.rstrip("\n").split("\t")
        if len(p) < 4 or p[1] != "0": continue
        d = p[2]
        m = re.search(r"HW=(\d+) N=(\d+) K=(\d+)", d)
        if not m: continue
        HW, N, K = (int(x) for x in m.groups()); M = rows * HW
        if d.startswith("conv3x3"):
            am = int(re.search(r"amode=(\d)", d).group(1)); cin = K // 9
            a_b = M * cin * 2 * (4 if am == 2 else 0.25 if am == 3 else 1)
        else:
            a_b = M * K * 2
        out_b = M * (N // 2 if d.startswith("geglu") else N) * 2
        tot += a_b + N * K * 2 + out_b + (M * N * 2 if "+res" in d else 0); n += 1
    return tot, n

ap = argparse.ArgumentParser()
for k in ("fetch", "write", "sq", "detail", "out"): ap.add_argument("--" + k, required=True)
ap.add_argument("--rows", type=int, required=True); ap.add_argument("--note", default=""); ap.add_argument("--trace", default="")
a = ap.parse_args()
tr = collections.defaultdict(lambda: [0, 0.0])       # family -> [calls, total ns] from *kernel_stats.csv
if a.trace:
    for f in glob.glob(os.path.join(a.trace, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Name"]); tr[fam][0] += int(r["Calls"]); tr[fam][1] += float(r["TotalDurationNs"])
fe, fe_d = load(a.fetch); wr, _ = load(a.write); sq, sq_d = load(a.sq)
out = {"note": a.note, "rows": a.rows}
for fam in sorted(set(fe) | set(wr) | set(sq)):
    o = {}
    if "FETCH_SIZE" in fe[fam]:
        n, v = fe[fam]["FETCH_SIZE"]; o["launches"] = n; o["fetch_bytes_per_launch"] = round(v * 1024 * 2.0 / n)
    if "WRITE_SIZE" in wr[fam]:
        n, v = wr[fam]["WRITE_SIZE"]; o["write_bytes_per_launch"] = round(v * 1024 / n)
    if "fetch_bytes_per_launch" in o and "write_bytes_per_launch" in o:
        o["hbm_bytes_per_launch"] = o["fetch_bytes_per_launch"] + o["write_bytes_per_launch"]
    if fam in tr and tr[fam][0]:
        o["avg_launch_us"] = round(tr[fam][1] / tr[fam][0] / 1e3, 2)        # rocprofv3 --kernel-trace --stats of the same command
        if "hbm_bytes_per_launch" in o: o["hbm_GBps"] = round(o["hbm_bytes_per_launch"] / (tr[fam][1] / tr[fam][0]), 1)
    s = sq[fam]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "SQ_BUSY_CU_CYCLES" in s and s["SQ_BUSY_CU_CYCLES"][1] > 0:
        o["mfma_util"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4.0 * s["SQ_BUSY_CU_CYCLES"][1]), 4)
    if "SQ_WAVE_CYCLES" in s and s["SQ_WAVE_CYCLES"][1] > 0:
        w = s["SQ_WAVE_CYCLES"][1]
        o["wave_time_split"] = {k2: round(s[k][1] / w, 3) for k, k2 in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "parked")) if k in s}
    out[fam] = o
if "igemm" in out:
    tot, n = algorithmic_bytes(a.detail, a.rows)
    out["igemm"]["algorithmic_bytes_per_launch"] = round(tot / max(n, 1)); out["igemm"]["launches_per_forward"] = n
os.makedirs(os.path.dirname(a.out), exist_ok=True)
json.dump(out, open(a.out, "w"), indent=1); print(json.dumps(out, indent=1))
