"""Attention kernel alone, hot caches: three vs four workgroups per CU (cfgpp_attention_set_occupancy) at the UNets' self-attention
shapes; correctness of both against torch SDPA in fp32."""
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
        H.lib().cfgpp_attention_set_occupancy(occ)
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
H.lib().cfgpp_attention_set_occupancy(4)
