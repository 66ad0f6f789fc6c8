"""Attention kernel alone, hot caches, at the UNets' self-attention shapes: the flash loop (cfgpp_attention_set_dma(1): four
workgroups per CU, softmax as a dependency chain inside every wave) against the woven kernel (mode 3: QK^T of tile t+1 and PV of
tile t-1 issued between slices of tile t's softmax, two workgroups per CU); correctness against torch SDPA in fp32 and against an
fp64 reference on a case that forces the re-reference branch.  Interleaved repeats; prints time, TFLOP/s and rel-L2 per mode."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
L = H.lib()
# the rare branch first (a wrong kernel should fail here, cheaply)
g = torch.Generator().manual_seed(77)
B, h, N, d = 1, 1, 1024, 64
q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
k[0, 0, 900] = q[0, 0, 5] * 4.0
k[0, 0, 130] = q[0, 0, 700] * 3.0
s = (q.double() @ k.double().transpose(-1, -2)) / d ** 0.5
ref = (torch.softmax(s, -1) @ v.double()).transpose(1, 2).reshape(B, N, h * d).float()
hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
for mode in (1, 3):
    L.cfgpp_attention_set_dma(mode)
    st = H.err_stats(H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp), ref)
    print(f"re-reference case, mode {mode}: {st}", flush=True)
for (B, h, N, d, what) in ((16, 8, 4096, 40, "SD1.5 64x64 level, 16 rows"), (4, 10, 4096, 64, "SDXL 64x64 level, 4 rows"), (4, 20, 1024, 64, "SDXL 32x32 level, 4 rows"),
                           (16, 20, 1024, 64, "SDXL 32x32 level, 16 rows"), (2, 8, 4096, 40, "SD1.5 64x64 level, 2 rows"), (1, 2, 256, 64, "4 key tiles")):
    g = torch.Generator().manual_seed(d + N)
    q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
    q = q * 1.5
    nb = min(B, 2)
    ref = F.scaled_dot_product_attention(q[:nb], k[:nb], v[:nb]).transpose(1, 2).reshape(nb, N, h * d)
    hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
    flops = 4.0 * B * h * N * N * d
    line = f"{what}: B*heads={B * h} N={N} d={d}"
    outs = {}
    for mode in (1, 3, 1, 3):
        L.cfgpp_attention_set_dma(mode)
        out = H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        outs[mode] = out
        err = float((out[:nb].float().cpu() - ref).norm() / ref.norm())
        for _ in range(3):
            H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        line += f" | mode {mode}: {us:7.1f} us {flops / us / 1e6:6.1f} TF/s rel {err:.1e}"
    line += f" | max |woven - flash| = {float((outs[3].float() - outs[1].float()).abs().max()):.2e}"
    print(line, flush=True)
L.cfgpp_attention_set_dma(1)
