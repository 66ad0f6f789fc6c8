"""Where the woven attention kernel's step time goes: the kernel with parts switched OFF (wrong results, timing only) -
cfgpp_attention_set_stagger(mask) in mode 3: 1 = no DMA / vmcnt waits, 2 = no barriers, 4 = no LDS fragment reads, 8 = no softmax VALU;
100 + n = the real kernel with LDS fragment reads issued n slots ahead of their MFMA (default 2)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
L = H.lib()
def t_us(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, h, N, d) in ((16, 8, 4096, 64), (4, 10, 4096, 64), (16, 20, 1024, 64)):
    g = torch.Generator().manual_seed(d + N)
    q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
    hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
    flops = 4.0 * B * h * N * N * d
    fn = lambda: H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
    L.cfgpp_attention_set_dma(1); L.cfgpp_attention_set_stagger(0)
    us = t_us(fn); line = f"B*heads={B * h} N={N} d={d}: flash loop {us:7.1f} us ({flops / us / 1e6:5.0f} TF/s)"
    L.cfgpp_attention_set_dma(3)
    for mask in (0, 103, 104, 106, 0, 103, 104, 106, 8, 15):
        L.cfgpp_attention_set_stagger(mask)
        us = t_us(fn); line += f" | woven -{mask}: {us:7.1f} ({flops / us / 1e6:5.0f})"
    print(line, flush=True)
L.cfgpp_attention_set_dma(1); L.cfgpp_attention_set_stagger(0)
