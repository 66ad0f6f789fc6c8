"""Host-side AddressSanitizer pass over the sequence that preceded the round-5 driver abort, in ONE process:
real SD1.5 engine at 16 rows (create, load, finalize, in-situ tile tuning incl. the K-split 8x8 level, forward, export of the
pins, profile with the detail buffer, destroy) -> 10-GB SDXL weight synthesis -> real SDXL engine at 4 / 2 / 16 rows -> destroy,
then the VAE and the graph-replay path.  Run through scripts/r06_runs/asan_repro.sh (LD_PRELOAD of the ASan runtime,
CFGPP_LIB = the host-instrumented twin library from `python -m cfgpp_amd.build --asan`)."""
import gc
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import realsize as RS  # noqa: E402
from cfgpp_amd import _lib  # noqa: E402
from cfgpp_amd.engine import HipUNet  # noqa: E402
from cfgpp_amd.unet_config import SD15, SDXL, TINY_SD  # noqa: E402
from cfgpp_amd.weights import synth_state_dict, synth_state_dict_iter  # noqa: E402

print("library:", _lib.LIB_PATH, _lib.build_id(), flush=True)
t0 = time.time()
i = RS.sd15_fwd_inputs()
net = HipUNet(SD15, max_rows=16, sample_hw=(64, 64))
net.load_state_dict(synth_state_dict_iter(SD15, 0)).finalize()
net.set_context(i["ehs"])
got = net.forward(i["z"].cuda(), 501.0).float().cpu()
pins = net.export_tuning()
pr = net.profile(i["z"].cuda(), 501.0, detail=True)
net.import_tuning(pins, 16)
got2 = net.forward(i["z"].cuda(), 501.0).float().cpu()
torch.cuda.synchronize()
print(f"sd15 16 rows: {len(pins)} pins, rel-L2 vs fixture {RS.rel_l2(got, RS.load_fixture('sd15_fwd')['eps'].float()):.2e}, "
      f"repeat identical {bool(torch.equal(got, got2))}, {time.time() - t0:.0f} s", flush=True)
del net
gc.collect()
t0 = time.time()
sd = synth_state_dict(SDXL, 0)                   # the allocation pattern of the test that died (a 10-GB dict)
print(f"sdxl weights: {sum(v.numel() for v in sd.values()) / 1e9:.2f} G params in {time.time() - t0:.0f} s", flush=True)
i = RS.sdxl_fwd_inputs()
net = HipUNet(SDXL, max_rows=16, sample_hw=(128, 128))
net.load_state_dict(sd).finalize()
del sd
gc.collect()
gold = RS.load_fixture("sdxl_fwd")["eps"].float()
for R, (zi, ci) in RS.SDXL_PLANS.items():
    net.set_context(i["ehs"][ci], i["te"][ci], i["ti"][ci])
    got = net.forward(i["z"][zi].cuda(), 501.0).float().cpu()
    print(f"sdxl rows {R}: rel-L2 {RS.rel_l2(got, gold[ci]):.2e}", flush=True)
del net
gc.collect()
# solver path on a small net: pin cache object, step kernels, graph replay, VAE decode + encode
os.environ["CFGPP_GRAPH"] = "1"
from cfgpp_amd.latent_diffusion import get_solver  # noqa: E402
s = get_solver("ddim_inversion_cfg++", solver_config=types.SimpleNamespace(num_sampling=5), device="cuda", unet_config=TINY_SD, max_batch=2)
uc, c = s.get_text_embed("bad", ["a cat", "a dog"])
img = torch.rand((2, 3, 128, 128)) * 2 - 1
out = s.sample(src_img=img, cfg_guidance=0.6, prompt_embeds=(uc, c))
print("tiny invert + regenerate + decode through graph replay:", tuple(out.shape), bool(torch.isfinite(out).all()), flush=True)
del s
gc.collect()
torch.cuda.synchronize()
print("ASAN_REPRO_DONE", flush=True)
