"""Average shader clock over back-to-back UNet forwards: clock_probe (s_memtime / s_memrealtime) before and after N predict() calls
on the engine's stream.   python scripts/r04_runs/forward_clock.py sd15 8 /tmp/libclock_probe.so"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cfgpp_amd.hip_engine import HipEngine
from cfgpp_amd.unet_config import CONFIGS
from cfgpp_amd.weights import synth_state_dict
name, B, so = sys.argv[1], int(sys.argv[2]), sys.argv[3]
probe = ctypes.CDLL(so).clock_probe; probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; probe.restype = None
cfg = CONFIGS[name]
cache = f"/tmp/cfgpp_synth_{name}.safetensors"
if not os.path.exists(cache):
    from safetensors.torch import save_file
    save_file({k: v.half().contiguous() for k, v in synth_state_dict(cfg, 0).items()}, cache + ".tmp"); os.replace(cache + ".tmp", cache)
eng = HipEngine(name, max_batch=B, weights=cache)
g = torch.Generator().manual_seed(0)
uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
te = ti = None
if cfg.addition_embed:
    te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
    ti = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B)).cuda()
eng.set_context(uc, c, te, ti)
z = torch.randn(B, 4, eng.H, eng.W, generator=g).cuda()
for _ in range(3): eng.predict(z, 500.0)
torch.cuda.synchronize()
buf = torch.zeros(4, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def clock_over(fn, label):
    torch.cuda.synchronize()
    probe(buf.data_ptr(), st)
    fn()
    probe(buf.data_ptr() + 16, st)
    torch.cuda.synchronize()
    a = buf.cpu().tolist()
    dt_us = (a[3] - a[1]) / 100.0
    print(f"{name} B={B} {label}: {dt_us / 1e3:.2f} ms, shader ticks / real time = {(a[2] - a[0]) / dt_us / 1e3:.3f} GHz", flush=True)
clock_over(lambda: time.sleep(0.05), "idle 50 ms (host sleep)")
for n in (1, 20, 100):
    clock_over(lambda: [eng.predict(z, 500.0) for _ in range(n)], f"{n} forward(s)")
clock_over(lambda: [eng.predict(z, 500.0) for _ in range(20)], "20 forwards again")
