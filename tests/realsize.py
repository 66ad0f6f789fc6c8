"""The real-size parity cases (real SD1.5 / SDXL nets at the benchmark's own sizes) - ONE definition of the inputs, used by

* ``tests/golden/make_unet_golden.py`` (build container, CPU): runs ``oracle/`` ONCE per case and commits the expected
  outputs as fixtures ``tests/golden/realsize_<case>.npz``;
* ``tests/test_gpu_realsize.py`` (-m gpu): starts ``python tests/realsize.py <case>`` as a SUBPROCESS per case - the HIP
  side of the case below runs there, is compared with the fixture, and prints one JSON line.  A native abort inside the
  library then fails that one test with its stderr instead of taking the pytest process (and every later test) with it,
  and the GPU box spends no minutes and no tens of GB on a CPU oracle.

Every input is seeded on the CPU generator, weights are ``cfgpp_amd.weights.synth_*`` (seed 0), text embeddings the
deterministic synthetic encoder's: both sides build bit-identical inputs from this file alone.  References:
the UNet call latent_diffusion.py:155 / latent_sdxl.py:181, the loops latent_diffusion.py:653-674, latent_sdxl.py:730-752,
838-858.
"""
from __future__ import annotations

import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
TVAL = 501.0


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().float()


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def fixture_path(case):
    return os.path.join(GOLDEN_DIR, f"realsize_{case}.npz")


def load_fixture(case):
    with np.load(fixture_path(case)) as f:
        return {k: torch.from_numpy(f[k]) for k in f.files}


# ---------------------------------------------------------------------------------------------------- inputs
def sd15_fwd_inputs():
    """C2's forward: 16 UNet rows (8 latents x {uc, c}) @ 64 x 64, 16 distinct contexts"""
    return dict(z=rnd(8, 4, 64, 64, seed=50), ehs=rnd(16, 77, 768, scale=0.5, seed=51))


def sdxl_fwd_inputs():
    """4 distinct oracle rows (2 latents x 2 contexts each) @ 128 x 128; the 2- / 4- / 16-row plans are built from them"""
    hw = 128
    return dict(z=rnd(2, 4, hw, hw, seed=50), ehs=rnd(4, 77, 2048, scale=0.5, seed=51), te=rnd(4, 1280, scale=0.5, seed=52),
                ti=torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * 4))


def small_fwd_inputs(cfg, R, hw):
    """the real nets at small row counts / other latent sizes (what tests/gpu_diag.py::unet_case feeds): SD1.5 2 rows @ 64 x 64,
    SDXL 2 rows @ 32 x 32 latents; two timesteps (981 and 1: both ends of the sinusoid)"""
    d = dict(z=rnd(R // 2, 4, hw, hw, seed=50), ehs=rnd(R, 77, cfg.cross_attention_dim, scale=0.5, seed=51), te=None, ti=None)
    if cfg.addition_embed:
        d["te"] = rnd(R, cfg.addition_pooled_dim, scale=0.5, seed=52)
        d["ti"] = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * R)
    return d


SMALL_FWD = {"sd15_fwd_r2": ("sd15", 2, 64), "sdxl_fwd_r2_32": ("sdxl", 2, 32)}
SMALL_TVALS = (981.0, 1.0)


# oracle row j of the SDXL forward: latent j % 2, context j.  plan rows -> (latent index per z row, context index per UNet row)
SDXL_PLANS = {4: ([0, 1], [0, 1, 2, 3]),
              2: ([0], [0, 2]),
              16: ([0, 1] * 4, [(r % 2) + 2 * ((r // 2) % 2) for r in range(16)])}

SD15_CHAIN = dict(name="ddim_cfg++", B=8, nfe=4, lam=0.6, seeds=list(range(100, 108)), prompts=[f"prompt {i}" for i in range(8)])
SDXL_CHAINS = [dict(name="ddim_cfg++", nfe=2, lam=0.6, B=2), dict(name="ddim_cfg++_lightning", nfe=1, lam=1.0, B=1)]
SDXL_PROMPTS = ["a cat wearing a hat", "a dog on a skateboard"]
SDXL_SEEDS = [31, 32]


def _sdxl_kw(leg):
    return dict(cfg_guidance=leg["lam"], target_size=(1024, 1024), original_size=(1024, 1024), seeds=SDXL_SEEDS[:leg["B"]],
                return_latents=True)


# ---------------------------------------------------------------------------------------------------- oracle side (CPU)
def oracle_sd15_fwd():
    from cfgpp_amd.unet_config import SD15
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    i = sd15_fwd_inputs()
    eps = UNetRef(SD15, synth_state_dict(SD15, 0))(torch.cat([i["z"], i["z"]]), TVAL, i["ehs"])["sample"]
    return dict(eps=eps.half().numpy())


def oracle_sdxl_fwd():
    from cfgpp_amd.unet_config import SDXL
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    i = sdxl_fwd_inputs()
    eps = UNetRef(SDXL, synth_state_dict(SDXL, 0))(torch.cat([i["z"], i["z"]]), TVAL, i["ehs"],
                                                   {"text_embeds": i["te"], "time_ids": i["ti"]})["sample"]
    return dict(eps=eps.half().numpy())


def _oracle_small_fwd(case):
    from cfgpp_amd.unet_config import CONFIGS
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    name, R, hw = SMALL_FWD[case]
    cfg = CONFIGS[name]
    i = small_fwd_inputs(cfg, R, hw)
    net = UNetRef(cfg, synth_state_dict(cfg, 0))
    ack = {"text_embeds": i["te"], "time_ids": i["ti"]} if cfg.addition_embed else None
    return {f"t{int(t)}": net(torch.cat([i["z"], i["z"]]), t, i["ehs"], ack)["sample"].half().numpy() for t in SMALL_TVALS}


def oracle_sd15_chain():
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import SD15
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    c = SD15_CHAIN
    net = UNetRef(SD15, synth_state_dict(SD15, 0))
    ref = get_solver(c["name"], solver_config=types.SimpleNamespace(num_sampling=c["nfe"]), device="cpu", max_batch=c["B"],
                     engine=MockEngine(lambda z, t, ehs, te, ti: net(z.float(), t, ehs.float())["sample"].half(), (64, 64)),
                     scalar_semantics="cuda")
    uc, cc = ref.get_text_embed("bad", c["prompts"])
    z0 = ref.sample(cfg_guidance=c["lam"], prompt_embeds=(uc, cc), seeds=c["seeds"], return_latents=True)[0]
    return dict(z0t=z0.float().numpy())


def oracle_sdxl_chain():
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.unet_config import SDXL
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    hw = 128
    net = UNetRef(SDXL, synth_state_dict(SDXL, 0))

    def unet(z, t, ehs, te, ti):
        return net(z.float(), t, ehs.float(), {"text_embeds": te.float(), "time_ids": ti.float()})["sample"].half()
    out = {}
    for leg in SDXL_CHAINS:
        ref = get_solver(leg["name"], solver_config=types.SimpleNamespace(num_sampling=leg["nfe"]), device="cpu", max_batch=leg["B"],
                         latent_hw=(hw, hw), engine=MockEngine(unet, (hw, hw)), scalar_semantics="cuda")
        p = SDXL_PROMPTS[:leg["B"]]
        pe = ref.get_text_embed("bad", p, "bad", p)
        out[leg["name"]] = ref.sample(prompt_embeds=pe, **_sdxl_kw(leg)).float().numpy()
    return out


ORACLE = {"sd15_fwd": oracle_sd15_fwd, "sdxl_fwd": oracle_sdxl_fwd, "sd15_chain": oracle_sd15_chain, "sdxl_chain": oracle_sdxl_chain,
          "sd15_fwd_r2": lambda: _oracle_small_fwd("sd15_fwd_r2"), "sdxl_fwd_r2_32": lambda: _oracle_small_fwd("sdxl_fwd_r2_32")}


# ---------------------------------------------------------------------------------------------------- HIP side (GPU box)
def cached_weights(cfg, name):
    """the seed-0 synthetic weights of `cfg` as (key, fp16 tensor) pairs.  Generating 2.6 G SDXL parameters takes a minute of one
    CPU core; the values are fp16-exact by construction (weights.synth_tensor), so the first case that needs them leaves an fp16
    copy under $TMPDIR and the other cases (own processes) read it back - lossless, and never part of the repo."""
    import tempfile
    from cfgpp_amd.weights import synth_state_dict_iter
    path = os.path.join(os.environ.get("CFGPP_WEIGHT_CACHE", tempfile.gettempdir()), f"cfgpp_synth_{name}_seed0_fp16.pt")
    if os.path.exists(path):
        try:
            return list(torch.load(path).items())
        except Exception:  # noqa: BLE001  (a torn file of a killed run)
            pass
    items = [(k, v.half()) for k, v in synth_state_dict_iter(cfg, 0)]
    try:
        tmp = f"{path}.{os.getpid()}.tmp"
        torch.save(dict(items), tmp)
        os.replace(tmp, path)
    except OSError:
        pass
    return items


def _build_id():
    from cfgpp_amd import _lib
    return _lib.build_id()


def hip_sd15_fwd():
    """autotune ON: 256-wide tiles, the K-split 8x8 level, d = 40 attention at N = 4096"""
    from cfgpp_amd.engine import HipUNet
    from cfgpp_amd.unet_config import SD15
    i, gold = sd15_fwd_inputs(), load_fixture("sd15_fwd")["eps"].float()
    net = HipUNet(SD15, max_rows=16, sample_hw=(64, 64))
    net.load_state_dict(cached_weights(SD15, "sd15")).finalize()
    net.set_context(i["ehs"])
    got = net.forward(i["z"].cuda(), TVAL).float().cpu()
    again = net.forward(i["z"].cuda(), TVAL).float().cpu()            # the tuned plan, second use
    torch.cuda.synchronize()
    rel = rel_l2(got, gold)
    worst = max(rel_l2(got[r], gold[r]) for r in range(16))
    ok = bool(torch.isfinite(got).all()) and rel < 2.5e-3 and worst < 4e-3 and bool(torch.equal(got, again))
    return dict(ok=ok, rel_l2=rel, worst_row=worst, repeat_bit_identical=bool(torch.equal(got, again)), tol=2.5e-3)


def hip_sdxl_fwd():
    """the real SDXL net at the row counts of all three SDXL workloads: 4 (C3, batch 2 per GPU), 2 (C5 edit, the K-split
    rule at the 32 x 32 level) and 16 (C4, Lightning batch 8)"""
    from cfgpp_amd.engine import HipUNet
    from cfgpp_amd.unet_config import SDXL
    i, gold = sdxl_fwd_inputs(), load_fixture("sdxl_fwd")["eps"].float()
    net = HipUNet(SDXL, max_rows=16, sample_hw=(128, 128))
    net.load_state_dict(cached_weights(SDXL, "sdxl")).finalize()
    res, ok = {}, True
    for R, (zi, ci) in SDXL_PLANS.items():
        net.set_context(i["ehs"][ci], i["te"][ci], i["ti"][ci])
        got = net.forward(i["z"][zi].cuda(), TVAL).float().cpu()
        rel = rel_l2(got, gold[ci])
        worst = max(rel_l2(got[r], gold[ci[r]]) for r in range(R))
        res[f"rows{R}"] = dict(rel_l2=rel, worst_row=worst)
        ok = ok and bool(torch.isfinite(got).all()) and rel < 2.5e-3 and worst < 4e-3          # measured 1.1e-3 at 4 rows
    return dict(ok=ok, tol=2.5e-3, **res)


def _hip_small_fwd(case):
    from cfgpp_amd.engine import HipUNet
    from cfgpp_amd.unet_config import CONFIGS
    name, R, hw = SMALL_FWD[case]
    cfg = CONFIGS[name]
    i, gold = small_fwd_inputs(cfg, R, hw), load_fixture(case)
    net = HipUNet(cfg, max_rows=R, sample_hw=(hw, hw))
    net.load_state_dict(cached_weights(cfg, name)).finalize()
    net.set_context(i["ehs"], i["te"], i["ti"])
    res, ok = {}, True
    for t in SMALL_TVALS:
        got = net.forward(i["z"].cuda(), t).float().cpu()
        rel = rel_l2(got, gold[f"t{int(t)}"].float())
        res[f"t{int(t)}"] = rel
        ok = ok and bool(torch.isfinite(got).all()) and rel < 2.5e-3          # tests/test_gpu_unet.py: EPS_REL
    return dict(ok=ok, tol=2.5e-3, **res)


def hip_sd15_chain():
    """4 NFE of the C2 job itself: real SD1.5 net, batch 8 -> 16 UNet rows, through get_solver + the fused step kernel"""
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import SD15
    c, gold = SD15_CHAIN, load_fixture("sd15_chain")["z0t"]
    hip = get_solver(c["name"], solver_config=types.SimpleNamespace(num_sampling=c["nfe"]), device="cuda", max_batch=c["B"],
                     unet_weights=iter(cached_weights(SD15, "sd15")))
    assert hip.scalar_semantics == "cuda"
    uc, cc = hip.get_text_embed("bad", c["prompts"])
    a = hip.sample(cfg_guidance=c["lam"], prompt_embeds=(uc, cc), seeds=c["seeds"], return_latents=True)[0].float().cpu()
    rel = rel_l2(a, gold)
    return dict(ok=bool(torch.isfinite(a).all()) and rel < 1e-3, rel_l2=rel, tol=1e-3)               # measured 4.1e-4


def hip_sdxl_chain():
    """C3: 2 NFE of ddim_cfg++ at batch 2; C4: 1 NFE of ddim_cfg++_lightning (lambda == 1: positive rows only, Q7)"""
    from cfgpp_amd.hip_engine import HipEngine
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.unet_config import SDXL
    gold = load_fixture("sdxl_chain")
    res, ok = {}, True
    eng = HipEngine(SDXL, max_batch=2, weights=iter(cached_weights(SDXL, "sdxl")))      # one engine serves both legs
    for leg in SDXL_CHAINS:
        hip = get_solver(leg["name"], solver_config=types.SimpleNamespace(num_sampling=leg["nfe"]), device="cuda", max_batch=leg["B"],
                         scalar_semantics="cuda", engine=eng)
        p = SDXL_PROMPTS[:leg["B"]]
        pe = hip.get_text_embed("bad", p, "bad", p)
        a = hip.sample(prompt_embeds=pe, **_sdxl_kw(leg))
        rows_seen = int(hip._ctx_keep[2].shape[0])
        rel = rel_l2(a, gold[leg["name"]])
        res[leg["name"]] = dict(rel_l2=rel, rows_seen=rows_seen)
        ok = (ok and tuple(a.shape) == (leg["B"], 4, 128, 128) and bool(torch.isfinite(a.float()).all()) and rel < 1.5e-3
              and rows_seen == (leg["B"] if leg["lam"] == 1.0 else 2 * leg["B"]))                    # measured 4.2e-4 / 5.9e-4
        del hip
    return dict(ok=ok, tol=1.5e-3, **res)


def hip_sd15_chain_graph():
    """the same 4-NFE batch-8 job as hipGraph replays of one captured step (cfgpp_sample_graph_ddim) on the real net: against the
    fixture, and bit-identical to the eager loop of the same solver"""
    os.environ["CFGPP_GRAPH"] = "0"
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import SD15
    c, gold = SD15_CHAIN, load_fixture("sd15_chain")["z0t"]
    hip = get_solver(c["name"], solver_config=types.SimpleNamespace(num_sampling=c["nfe"]), device="cuda", max_batch=c["B"],
                     unet_weights=iter(cached_weights(SD15, "sd15")))
    uc, cc = hip.get_text_embed("bad", c["prompts"])
    run = lambda: hip.sample(cfg_guidance=c["lam"], prompt_embeds=(uc, cc), seeds=c["seeds"], return_latents=True)[0].float().cpu()  # noqa: E731
    eager = run()
    os.environ["CFGPP_GRAPH"] = "1"
    graph, again = run(), run()
    rel = rel_l2(graph, gold)
    same = bool(torch.equal(eager, graph)) and bool(torch.equal(graph, again))
    return dict(ok=bool(torch.isfinite(graph).all()) and rel < 1e-3 and same, rel_l2=rel, graph_equals_eager=same, tol=1e-3)


HIP = {"sd15_fwd": hip_sd15_fwd, "sdxl_fwd": hip_sdxl_fwd, "sd15_chain": hip_sd15_chain, "sdxl_chain": hip_sdxl_chain,
       "sd15_chain_graph": hip_sd15_chain_graph,
       "sd15_fwd_r2": lambda: _hip_small_fwd("sd15_fwd_r2"), "sdxl_fwd_r2_32": lambda: _hip_small_fwd("sdxl_fwd_r2_32")}


def main(argv):
    case = argv[1]
    t0 = time.time()
    out = HIP[case]()
    torch.cuda.synchronize()
    out.update(case=case, seconds=round(time.time() - t0, 1), build_id=_build_id())
    print("REALSIZE_RESULT " + json.dumps(out), flush=True)
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
