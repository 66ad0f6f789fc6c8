"""rocprofv3 target / summariser for the attention kernels alone.
  run:        rocprofv3 --pmc ... -d DIR -o p --output-format csv -- python scripts/r04_runs/pmc_attention.py run
  summarise:  python scripts/r04_runs/pmc_attention.py sum DIR
Per kernel instantiation: launches, average duration, GRBM_GUI_ACTIVE per ns (the clock the chip actually ran at, if the counter is
per-device), matrix-pipe utilisation SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), VALU-issue share, wave time split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hip_ops as H
    L = H.lib()
    for (B, h, N, d) in ((16, 8, 4096, 64), (16, 8, 4096, 40), (4, 20, 1024, 64)):
        g = torch.Generator().manual_seed(d + N)
        q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
        hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
        for mode, masks in ((1, (0,)), (3, (0, 8, 15, 7, 1) if d == 64 else (0,))):
            L.cfgpp_attention_set_dma(mode)
            for mask in masks:
                L.cfgpp_attention_set_stagger(mask)
                for _ in range(8):
                    H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
                torch.cuda.synchronize()
    L.cfgpp_attention_set_dma(1); L.cfgpp_attention_set_stagger(0)
else:
    import csv, glob, collections, re
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(list); seen = set()
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "attn" not in k: continue
            m = re.search(r"(x?attn\w*_kernel)<([^>]*)>", k)
            key = (m.group(1) + "<" + m.group(2) + ">" if m else k[:60]) + " grid " + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"])); dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for key in agg:
        a = agg[key]; ds = sorted(dur[key]); n = len(ds); tot = sum(ds); med = ds[n // 2]
        cu = a.get("SQ_BUSY_CU_CYCLES", 0.0)
        line = f"{key}: launches {n} median {med / 1e3:.1f} us"
        if "GRBM_GUI_ACTIVE" in a: line += f" | GRBM_GUI_ACTIVE/ns {a['GRBM_GUI_ACTIVE'] / tot:.3f}"
        if cu:
            line += f" | busy-CU cycles/ns/256 {cu / tot / 256:.3f} | mfma_util {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4 * cu):.3f}"
            if "SQ_ACTIVE_INST_VALU" in a: line += f" | valu_active {a['SQ_ACTIVE_INST_VALU'] / (4 * cu):.3f}"
        w = a.get("SQ_WAVE_CYCLES", 0.0)
        if w: line += f" | waves: issuing {a.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} issue-stall {a.get('SQ_WAIT_INST_ANY', 0) / w:.3f}"
        if "SQ_INSTS_VALU" in a: line += f" | VALU insts/launch {a['SQ_INSTS_VALU'] / n:.3e}"
        print(line)
