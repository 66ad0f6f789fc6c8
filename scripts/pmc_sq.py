"""Aggregate rocprofv3 --pmc SQ counters per kernel: python scripts/pmc_sq.py <dir> [<dir> ...]"""
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k[k.find("igemm_kernel"):][:60] if "igemm_kernel" in k else k[:60]
            a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"    {c:32s} n={n:4d}  avg={v / n:16.1f}")
