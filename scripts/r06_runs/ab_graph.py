"""Same-box, same-process A/B of the whole sampling loop: eager launches vs hipGraph replay (cfgpp_sample_graph_ddim).
    python scripts/r06_runs/ab_graph.py sd15 8 50 [rounds]      |     ... sdxl 2 50
One solver / one engine; `CFGPP_GRAPH` is flipped between interleaved rounds (eager, graph, eager, graph ...); each round is
one full `sample(return_latents=True)` job timed wall-clock around a device sync - the number bench.py's ms_per_step is made of
(minus the VAE).  Also reports: host time spent inside sample() before the final sync (how long the Python thread is busy
enqueueing - what a graph frees), and that both modes return bit-identical latents."""
import os
import statistics
import sys
import time
import types

os.environ.setdefault("CFGPP_TUNE_CACHE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

name, B, nfe = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sc = types.SimpleNamespace(num_sampling=nfe)
if name == "sd15":
    from cfgpp_amd.latent_diffusion import get_solver
    s = get_solver("ddim_cfg++", solver_config=sc, device="cuda", max_batch=B)
    uc, c = s.get_text_embed("bad", [f"prompt {i}" for i in range(B)])
    run = lambda: s.sample(cfg_guidance=0.6, prompt_embeds=(uc, c), seeds=list(range(B)), return_latents=True)[0]  # noqa: E731
else:
    from cfgpp_amd.latent_sdxl import get_solver
    s = get_solver("ddim_cfg++", solver_config=sc, device="cuda", max_batch=B)
    p = [f"prompt {i}" for i in range(B)]
    pe = s.get_text_embed("bad", p, "bad", p)
    run = lambda: s.sample(prompt_embeds=pe, cfg_guidance=0.6, target_size=(1024, 1024), original_size=(1024, 1024),  # noqa: E731
                           seeds=list(range(B)), return_latents=True)
from cfgpp_amd import _lib  # noqa: E402
print(f"# {name} B={B} nfe={nfe} build {_lib.build_id()} device {torch.cuda.get_device_name(0)}", flush=True)
res = {"0": [], "1": []}
host = {"0": [], "1": []}
outs = {}
for mode in ("0", "1"):                      # warm both paths: tuning, capture
    os.environ["CFGPP_GRAPH"] = mode
    outs[mode] = run().clone()
    torch.cuda.synchronize()
print("bit-identical latents (eager vs graph):", bool(torch.equal(outs["0"], outs["1"])), flush=True)
for r in range(rounds):
    for mode in ("0", "1"):
        os.environ["CFGPP_GRAPH"] = mode
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z = run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[mode].append((t2 - t0) * 1e3)
        host[mode].append((t1 - t0) * 1e3)
        assert torch.equal(z, outs[mode])
for mode, label in (("0", "eager"), ("1", "graph")):
    print(f"{label}: job wall ms min {min(res[mode]):.2f} median {statistics.median(res[mode]):.2f}  ({min(res[mode]) / nfe:.3f} ms/step)   "
          f"host busy ms median {statistics.median(host[mode]):.2f}   all: " + " ".join(f"{x:.1f}" for x in res[mode]), flush=True)
d = (min(res["1"]) - min(res["0"])) / min(res["0"]) * 100
print(f"graph vs eager: {d:+.2f} % job wall (min of {rounds})")
