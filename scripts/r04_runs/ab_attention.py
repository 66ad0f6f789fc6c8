"""Attention kernel alone, hot caches, at the UNets' self-attention shapes; correctness against torch SDPA in fp32.  Calls 9 / 10 of
round 4 ran it on the builds that still carried both forms (three workgroups per CU on a 3-stage ring vs four on a 2-stage ring,
switch cfgpp_attention_set_occupancy; profiles/r04/ab/attention_occupancy_call9.txt, _call10.txt); on a build without the switch it
times the shipped form twice per "occ" column."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
for (B, h, N, d, what) in ((16, 8, 4096, 40, "SD1.5 64x64 level, 16 rows"), (4, 10, 4096, 64, "SDXL 64x64 level, 4 rows"), (4, 20, 1024, 64, "SDXL 32x32 level, 4 rows"),
                           (16, 20, 1024, 64, "SDXL 32x32 level, 16 rows")):
    g = torch.Generator().manual_seed(d + N)
    q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
    ref = None
    if B * h * N <= 16 * 8 * 4096:
        ref = F.scaled_dot_product_attention(q[:2], k[:2], v[:2]).transpose(1, 2).reshape(2, N, h * d)
    hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
    flops = 4.0 * B * h * N * N * d
    line = f"{what}: B*heads={B * h} N={N} d={d}"
    for occ in (3, 4, 3, 4):
        try:
            H.lib().cfgpp_attention_set_occupancy(occ)
        except AttributeError:
            pass
        out = H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        err = float((out[:2].float().cpu() - ref).norm() / ref.norm()) if ref is not None else float("nan")
        for _ in range(3):
            H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" | occ {occ}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s rel {err:.1e}"
    print(line, flush=True)
try:
    H.lib().cfgpp_attention_set_occupancy(4)
except AttributeError:
    pass
