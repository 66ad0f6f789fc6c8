"""Run ONE attention shape in a loop (PMC / timing target): python scripts/one_attn.py B heads N d [Nk] [iters]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as H
from cfgpp_amd import _lib
B, h, N, d = (int(x) for x in sys.argv[1:5])
Nk = int(sys.argv[5]) if len(sys.argv) > 5 else N
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dp = H.round_up(d, 32)
qp, kp = H.round_up(N, 128), H.round_up(Nk, 64)
hq = torch.randn(B * h, qp, dp, device="cuda", dtype=torch.float16)
hk = torch.randn(B * h, kp, dp, device="cuda", dtype=torch.float16)
hvt = torch.randn(B * h, dp, kp, device="cuda", dtype=torch.float16)
if d % 32: hq[:, :, d:] = 0; hk[:, :, d:] = 0; hvt[:, d:, :] = 0
_lib.check(H.lib().cfgpp_op_attention_prepare_vt(H.P(hvt), B * h, d, kp, H.stream()), "prep")
o = torch.empty(B, N, h * d, device="cuda", dtype=torch.float16)
H.lib().cfgpp_attention_set_dma(int(os.environ.get("ATTN_MODE", "1")))
H.lib().cfgpp_attention_set_stagger(int(os.environ.get("ATTN_STAGGER", "0")))
H.lib().cfgpp_attention_set_cross(int(os.environ.get("ATTN_CROSS", "1")))
fn = lambda: _lib.check(H.lib().cfgpp_op_attention(H.P(hq), H.P(hk), H.P(hvt), H.P(o), B, h, d, N, Nk, qp, kp, H.stream()), "attn")
for _ in range(3): fn()
torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): fn()
e.record(); torch.cuda.synchronize()
dt = s.elapsed_time(e) / iters * 1e-3
print(f"attn B={B} h={h} N={N} Nk={Nk} d={d} mode={os.environ.get('ATTN_MODE', '1')} stagger={os.environ.get('ATTN_STAGGER', '0')} cross={os.environ.get('ATTN_CROSS', '1')}: {dt*1e6:.1f} us  {4.0*B*h*N*Nk*d/dt/1e12:.1f} TF/s", flush=True)
