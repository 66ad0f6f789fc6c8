#!/usr/bin/env python
"""Benchmark of the CFG++ sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config sd15|sdxl|sdxl_lightning|sdxl_edit]

One "step" = one pass of the hot path over one batch of synthetic prompts: B
independent 50-NFE DDIM-CFG++ chains (UNet at batch 2B + fused step kernel per
NFE) followed by the VAE decode and the device->host copy of the images, i.e.
exactly what ``solver.sample()`` returns in the reference
(latent_diffusion.py:634-679).  Default workload = BASELINE.json configs[1]:
SD1.5 512x512, ddim_cfg++, 50 NFE, lambda = 0.6, batch 8 on one MI355X.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs the
same per-GPU batch (weak scaling; ``--global-batch G`` fixes the TOTAL batch and gives
every rank G / N chains instead: strong scaling, BASELINE configs 3 - 5 as written) on its
own prompt shard; rank 0 broadcasts the conditioning once over RCCL; nothing is exchanged
inside the loop.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as cfgpp_amd/__init__.py (must precede the first HIP call)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_FP16 = 2.5e15      # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
VAE_DEC_FLOPS = {512: 2.51e12, 1024: 10.47e12}
VAE_ENC_FLOPS = {512: 1.12e12, 1024: 4.88e12}

WORKLOADS = {
    # name: (module, solver, cfg name, NFE, lambda, per-GPU batch, image size, description)
    "sd15": ("sd", "ddim_cfg++", "sd15", 50, 0.6, 8, 512, "SD1.5 512x512 ddim_cfg++ 50 NFE lambda=0.6 batch=8/GPU"),
    "sdxl": ("xl", "ddim_cfg++", "sdxl", 50, 0.6, 2, 1024, "SDXL 1024x1024 ddim_cfg++ 50 NFE lambda=0.6 batch=2/GPU"),
    "sdxl_lightning": ("xl", "ddim_cfg++_lightning", "sdxl", 4, 1.0, 8, 1024,
                       "SDXL-Lightning arch 1024x1024 ddim_cfg++_lightning 4 NFE lambda=1.0 batch=8/GPU"),
    "sdxl_edit": ("xl", "ddim_inversion_cfg++", "sdxl", 50, 0.6, 1, 1024,
                  "SDXL 1024x1024 ddim_edit_cfg++ (tgt=src) 50+50 NFE lambda=0.6 batch=1/GPU"),
}


_T0 = time.time()


def log(msg):
    if os.environ.get("CFGPP_BENCH_VERBOSE", "1") != "0":
        print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="sd15", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (weak scaling: fixed work per GPU)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: total batch of the job, split evenly over the --gpus ranks (BASELINE configs 3/4/5 are "
                         "global batches: --config sdxl --global-batch 16, sdxl_lightning 64, sdxl_edit 8 at 1/2/4/8 GPUs)")
    ap.add_argument("--nfe", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short SDXL 1024^2 leg the default (sd15) run appends")
    return ap.parse_args()


def make_solver(kind, name, cfg_name, nfe, batch, device):
    from cfgpp_amd.unet_config import CONFIGS
    cfg = CONFIGS[cfg_name]
    sc = types.SimpleNamespace(num_sampling=nfe)
    if kind == "sd":
        from cfgpp_amd.latent_diffusion import get_solver
    else:
        from cfgpp_amd.latent_sdxl import get_solver
    return get_solver(name, solver_config=sc, device=device, max_batch=batch), cfg


def prompts_for(lo, hi):
    base = ["a photo of an astronaut riding a horse on mars", "a watercolor painting of a lighthouse at dawn",
            "a bowl of ramen on a wooden table, studio lighting", "a red fox in a snowy forest",
            "an isometric render of a tiny city", "a portrait of an old sailor, oil on canvas",
            "a macro shot of a dew drop on a leaf", "a futuristic train crossing a desert"]
    return [f"{base[i % len(base)]} #{i}" for i in range(lo, hi)]


NULL = "low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,"


def cpu_baseline_child(cfg_name, name, nfe, lam, img):
    """Runs in a child process (so the parent can bound it): the CPU restatement (oracle/) of the same
    path on the host cores - a bounded sample (UNet batch-2 + step iterations of ONE chain, then one VAE
    decode), printed as JSON lines as soon as each part is measured."""
    from cfgpp_amd.schedule import SchedulerTables
    from cfgpp_amd.unet_config import CONFIGS
    from oracle.vae_ref import VAERef as TorchVAE
    from cfgpp_amd.weights import synth_state_dict
    from oracle import sampler as O
    from oracle.unet_ref import UNetRef
    cfg = CONFIGS[cfg_name]
    cores = os.cpu_count() or 1
    threads = min(cores, 64)                 # beyond ~64 threads torch-CPU conv/GEMM stops scaling on this host
    torch.set_num_threads(threads)
    net = UNetRef(cfg, synth_state_dict(cfg, 0))
    tb = SchedulerTables(nfe)
    hw = img // 8
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 4, hw, hw, generator=g)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.5
    ack = None
    if cfg.addition_embed:
        ack = {"text_embeds": torch.randn(2, cfg.addition_pooled_dim, generator=g) * 0.5,
               "time_ids": torch.tensor([[img, img, 0, 0, img, img]] * 2, dtype=torch.float32)}

    def unet(zz, t):
        eps = net(torch.cat([zz, zz]), float(t), ehs, ack)["sample"].half()
        return eps[:1], eps[1:]
    ts = tb.timesteps
    unet(z, ts[0])                          # warm-up (thread pool, allocator)
    n_iter = 2
    t0 = time.time()
    zt = z
    for t in ts[:n_iter]:
        eu, ec = unet(zt, t)
        _, zt = O.ddim_step(zt, eu, ec, lam, None, None, False, True, sqrt4=tb.ddim_sqrt_coeffs(t))
    s_per_iter = (time.time() - t0) / n_iter
    n_unet = nfe * (2 if "inversion" in name else 1)
    out = {"value": round(1.0 / (n_unet * s_per_iter), 6), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": f"{n_iter} UNet(batch 2)+fused-step iterations of one chain at {hw}x{hw} timed ({s_per_iter:.2f} s/iter), "
                     f"extrapolated to {n_unet} iterations; VAE decode not yet included"}
    print(json.dumps(out), flush=True)
    vae = TorchVAE(cfg.vae_scale, device="cpu", dtype=torch.float32)
    t0 = time.time()
    vae.decode(zt)
    dec_s = time.time() - t0
    out["value"] = round(1.0 / (n_unet * s_per_iter + dec_s), 6)
    out["sample"] = (f"{n_iter} UNet(batch 2)+fused-step iterations of one chain at {hw}x{hw} ({s_per_iter:.2f} s/iter) and one VAE "
                     f"decode ({dec_s:.2f} s) timed on {threads} threads, extrapolated to {n_unet} iterations + decode")
    print(json.dumps(out), flush=True)


def cpu_baseline(cfg_name, name, nfe, lam, img, limit_s=240):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", cfg_name, name, str(nfe), str(lam), str(img)]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    try:
        outp, _ = p.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        p.kill()
        outp, _ = p.communicate()
    lines = [ln for ln in (outp or "").strip().splitlines() if ln.startswith("{")]
    if not lines:
        return {"error": f"cpu baseline produced nothing within {limit_s}s"}
    return json.loads(lines[-1])


PROFILE_ROUND = "r06"


def roofline_block(eng, config, B, dev, rnd):
    """Dominant kernel = the implicit-GEMM conv/linear kernel.  `achieved` = algorithmic FLOPs of its launches in one
    UNet forward / their summed HIP-event durations on the launch stream (`cfgpp_unet_profile`: three profiled forwards
    right after the timed region; the MEDIAN pass's family sum is `achieved`, the per-launch minimum over the three passes is
    reported beside it as `achieved_per_launch_min`; all three per-family sums are in the JSON line).  Live as well: the wall
    time of whole forwards at the job's batch (`unet_forward_wall_ms`), attention TFLOP/s, and algorithmic GB/s of the HBM-bound families
    (GroupNorm / LayerNorm at 4 B per element, the fused CFG++ step at 16 B per latent element).  From the committed
    rocprofv3 PMC passes of the SAME population (UNet-only forwards at this batch with the tiles this build's tuner
    pins; scripts/pmc_unet.py + scripts/pmc_summary.py -> profiles/<round>/pmc_<config>_b<B>.json): `traffic` (HBM
    bytes per igemm launch, FETCH_SIZE x2 + WRITE_SIZE per the gfx950 note of the guide) and `mfma_util`
    (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)) of the igemm and attention kernels."""
    import re
    # whole UNet forward at the job's batch, as the sampling loop runs it: wall time of back-to-back predict() calls
    zf = torch.randn((B, 4, eng.H, eng.W), device=dev)
    for _ in range(2):
        eng.predict(zf, 500.0)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(10):
        eng.predict(zf, 500.0)
    f1.record()
    torch.cuda.synchronize()
    forward_wall_ms = f0.elapsed_time(f1) / 10
    rows = 2 * B
    z = zf
    KIND = {"0": "igemm", "1": "attention", "2": "norm", "3": "small"}
    passes, flops, launches = [], {}, {}
    for t in (981.0, 501.0, 21.0):
        pr = eng.unet.profile(z, t, detail=True)
        for k in KIND.values():
            flops[k], launches[k] = pr[k]["flops"], pr[k]["launches"]
        passes.append([line.split("\t") for line in pr["detail"].strip().split("\n")])
    # Per launch: the FASTEST of the three profiled passes.  A launch's duration does not depend on t, but an interval
    # between two events also contains whatever kept the stream waiting before the launch was enqueued: the round-2
    # final run had single ~57 ms host stalls land inside one attention (SD1.5) / one igemm (SDXL) interval.
    fam = {k: dict(ms=0.0, flops=flops[k], launches=launches[k]) for k in KIND.values()}
    gb = {"groupnorm": [0.0, 0.0], "layernorm": [0.0, 0.0]}      # [bytes, seconds]
    pass_ms = [{k: 0.0 for k in KIND.values()} for _ in passes]
    for i, parts in enumerate(passes[0]):
        if len(parts) < 4:
            continue
        us_all = [float(p[i][3]) for p in passes if i < len(p) and len(p[i]) >= 4]
        us = min(us_all)
        kind = KIND.get(parts[1], "small")
        fam[kind]["ms"] += us * 1e-3
        for q, u in enumerate(us_all):
            pass_ms[q][kind] += u * 1e-3
        m = re.match(r"(groupnorm|layernorm) HW=(\d+) C=(\d+)", parts[2])
        if m:
            gb[m.group(1)][0] += 4.0 * rows * int(m.group(2)) * int(m.group(3))
            gb[m.group(1)][1] += us * 1e-6
    log("profiled passes, ms per family: " + " | ".join(", ".join(f"{k} {v:.2f}" for k, v in pm.items()) for pm in pass_ms)
        + " | per-launch minimum: " + ", ".join(f"{k} {v['ms']:.2f}" for k, v in fam.items()))
    # fused CFG++ step kernel: 16 B per latent element (4 z in, 2 + 2 eps in, 4 z0t out, 4 z out)
    from cfgpp_amd import engine as E
    zz = torch.randn((B, 4, eng.H, eng.W), device=dev)
    z0 = torch.empty_like(zz)
    eu, ec = (torch.randn((B, 4, eng.H, eng.W), device=dev).half() for _ in range(2))
    for _ in range(3):
        E.step_ddim(zz, z0, eu, ec, 0.6, (0.5, 0.8, 0.7, 0.6), False, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        E.step_ddim(zz, z0, eu, ec, 0.6, (0.5, 0.8, 0.7, 0.6), False, True)
    e1.record()
    torch.cuda.synchronize()
    step_s = e0.elapsed_time(e1) / 20 * 1e-3
    ig = fam["igemm"]
    # `achieved` = the MEDIAN of the three profiled passes' family sums (a forward that actually ran); the per-launch minimum
    # over the passes (an optimistic order statistic) is reported beside it
    med = {k: sorted(pm[k] for pm in pass_ms)[len(pass_ms) // 2] for k in KIND.values()}
    ach = ig["flops"] / (med["igemm"] * 1e-3) / 1e12
    ach_min = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
    pmc = None
    pf = os.path.join(ROOT, "profiles", rnd, f"pmc_{config}_b{B}.json")
    if os.path.exists(pf):
        try:
            pmc = json.load(open(pf))
        except Exception:  # noqa: BLE001
            pmc = None
    out = {"bound": "mfma", "kernel": "implicit-GEMM family: igemm_kernel / igemm16_kernel / tile32_kernel / big4_kernel / big4p_kernel (conv3x3, conv1x1, linear)", "achieved": round(ach, 1),
           "peak": PEAK_MFMA_FP16 / 1e12, "unit": "TFLOP/s", "frac": round(ach / (PEAK_MFMA_FP16 / 1e12), 4),
           "traffic": None if pmc is None else pmc["igemm"].get("hbm_bytes_per_launch"),
           "traffic_unit": "HBM bytes per igemm launch (rocprofv3 PMC passes over UNet-only forwards at this batch)",
           "algorithmic_bytes_per_launch": None if pmc is None else pmc["igemm"].get("algorithmic_bytes_per_launch"),
           "mfma_util": None if pmc is None else {k: pmc[k].get("mfma_util") for k in ("igemm", "attention") if k in pmc},
           "achieved_per_launch_min": round(ach_min, 1),
           "launches_per_forward": ig["launches"], "avg_launch_us": round(med["igemm"] / max(ig["launches"], 1) * 1e3, 2),
           "unet_forward_wall_ms": round(forward_wall_ms, 3), "unet_forward_rows": 2 * B,
           "unet_forward_TFLOPs": round(eng.flops_per_forward(2 * B) / (forward_wall_ms * 1e-3) / 1e12, 1),
           "per_family_ms_per_forward": {k: round(v, 3) for k, v in med.items()},
           "per_family_ms_per_launch_min": {k: round(v["ms"], 3) for k, v in fam.items()},
           "per_family_ms_per_profiled_pass": [{k: round(v, 3) for k, v in pm.items()} for pm in pass_ms],
           "attention_TFLOPs": round(fam["attention"]["flops"] / (med["attention"] * 1e-3) / 1e12, 1),
           "hbm_GBps": {"groupnorm": round(gb["groupnorm"][0] / max(gb["groupnorm"][1], 1e-12) / 1e9, 1),
                        "layernorm": round(gb["layernorm"][0] / max(gb["layernorm"][1], 1e-12) / 1e9, 1),
                        "cfgpp_step": round(16.0 * zz.numel() / step_s / 1e9, 1),
                        "note": "algorithmic bytes (GroupNorm / LayerNorm 4 B per element, step 16 B per latent element) / HIP-event time; "
                                "the step kernel moves %.0f KB per launch and is launch-latency bound" % (16.0 * zz.numel() / 1e3),
                        "peak": 8000.0},
           "pmc_source": None if pmc is None else os.path.relpath(pf, ROOT),
           "pmc_note": None if pmc is None else pmc.get("note")}
    return out


def prepare_job(solver, cfg, kind, name, B, img, lam, rank, world, dev):
    """The multi-GPU plumbing of one benchmark job, shared by main() and the world-2 gloo test: rank 0 "encodes" every
    prompt of the global batch, ONE packed broadcast of the conditioning (RCCL over xGMI on the GPU box), every rank
    keeps its contiguous shard of B prompts / seeds.  Returns (one_job, global_batch); ``one_job()`` runs the rank's B
    chains + decode and returns the images.  Nothing is exchanged inside the sampling loop."""
    from cfgpp_amd import dist as D
    total = B * world
    D_ = cfg.cross_attention_dim
    shapes = [((1, 77, D_), torch.float16), ((total, 77, D_), torch.float16)]
    if kind == "xl":
        shapes += [((1, cfg.addition_pooled_dim), torch.float16), ((total, cfg.addition_pooled_dim), torch.float16)]
    payload = [None] * len(shapes)
    if rank == 0:
        if kind == "sd":
            uc, c = solver.get_text_embed(NULL, prompts_for(0, total))
            payload = [uc, c]
        else:
            p = prompts_for(0, total)
            ne, pe, pn, pp = solver.get_text_embed(NULL, p, NULL, p)
            payload = [ne, pe, pn, pp]
    if torch.cuda.is_available() and torch.device(dev).type == "cuda":
        torch.cuda.synchronize()
    t_b = time.perf_counter()
    cond = D.broadcast_conditioning(payload, shapes, dev)
    if torch.cuda.is_available() and torch.device(dev).type == "cuda":
        torch.cuda.synchronize()
    prepare_job.broadcast_ms = (time.perf_counter() - t_b) * 1e3
    lo, hi = D.shard_range(total, rank, world)
    seeds = [42 + i for i in range(lo, hi)]
    src_img = None
    if "inversion" in name:      # seeded synthetic source images in [-1, 1] (per global prompt index); the VAE encode is inside the timed job
        imgs = []
        for i in range(lo, hi):
            g = torch.Generator().manual_seed(7 + i)
            imgs.append(torch.rand((1, 3, img, img), generator=g) * 2 - 1)
        src_img = torch.cat(imgs).to(dev)

    def one_job(**extra):
        # a real stream of jobs brings new prompts every time: drop the solver's conditioning cache so that set_context
        # (cross-attention K / V^T of every block, SDXL's added-condition embedding) runs INSIDE every timed job
        solver._ctx_key = None
        if kind == "sd":
            return solver.sample(cfg_guidance=lam, prompt=None, prompt_embeds=(cond[0], cond[1][lo:hi].contiguous()), seeds=seeds, **extra)
        pe = (cond[0], cond[1][lo:hi].contiguous(), cond[2], cond[3][lo:hi].contiguous())
        if "inversion" in name:
            pe = (pe[0], pe[1], pe[1], pe[2], pe[3], pe[3])
            return solver.sample(cfg_guidance=lam, prompt_embeds=pe, target_size=(img, img), src_img=src_img, **extra)
        return solver.sample(cfg_guidance=lam, prompt_embeds=pe, target_size=(img, img), seeds=seeds, **extra)
    return one_job, total


def run_workload(args, config, rank, world, dev, steps, warmup, batch=0, nfe_override=0, with_cpu_baseline=True, scaling="weak"):
    """build the engine of one WORKLOADS entry, share rank 0's tile pins, warm up, time `steps` jobs between barriers
    (max over ranks) and return the JSON fields of that workload"""
    from cfgpp_amd import dist as D
    kind, name, cfg_name, nfe, lam, B, img, desc = WORKLOADS[config]
    if batch and batch != B:
        import re
        desc = re.sub(r"batch=\d+/GPU", f"batch={batch}/GPU", desc)      # the line names the per-GPU batch that actually ran
    B = batch or B
    nfe = nfe_override or nfe
    log(f"building engine {cfg_name} max_batch={B}")
    solver, cfg = make_solver(kind, name, cfg_name, nfe, B, dev)
    eng = solver.engine
    log(f"engine ready, device memory {eng.device_bytes() / 1e9:.2f} GB")

    one_job, total = prepare_job(solver, cfg, kind, name, B, img, lam, rank, world, dev)
    broadcast_ms = getattr(prepare_job, "broadcast_ms", 0.0)
    tuning = "in situ on this GPU (first forward at this batch)"
    if world > 1:
        # rank 0 tunes (its first job); its pins go to every rank, so N ranks neither spend 28 tuning forwards each nor can pin
        # different tiles (every tuner candidate gives bit-identical results; the pins only decide speed)
        hints = None
        if rank == 0:
            one_job()
            sync(dev)
            hints = eng.export_tuning()
        hints = D.broadcast_ints(hints, dev)
        if rank != 0 and hints:
            eng.import_tuning(hints, B)
        tuning = f"rank 0's {len(hints)} pins broadcast to all ranks"
    for i in range(warmup):
        out = one_job()
        sync(dev)
        log(f"warmup job {i} done")
    D.barrier()
    sync(dev)
    job_s = []
    t0 = time.perf_counter()
    for _ in range(steps):
        tj = time.perf_counter()
        out = one_job()                      # returns host images: the D2H copy at its end is the job's own sync point
        job_s.append(time.perf_counter() - tj)
    sync(dev)
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    log(f"timed region done: {dt:.2f}s for {steps} jobs")
    assert out.shape == (B, 3, img, img) and bool(torch.isfinite(out).all())
    per_rank = D.gather_floats(job_s, dev)

    images = total * steps
    value = images / dt
    n_unet = nfe * (2 if "inversion" in name else 1)
    rows = 2 * B
    unet_flops = eng.flops_per_forward(rows)          # algorithmic, per forward at this batch
    flops_per_image = (n_unet * unet_flops / B) + VAE_DEC_FLOPS.get(img, 0.0) + (VAE_ENC_FLOPS.get(img, 0.0) if "inversion" in name else 0.0)
    flat = [x for r in per_rank for x in r]
    result = {
        "metric": "images/sec", "value": round(value, 4), "unit": "images/sec", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
        "config": {"workload": desc, "per_gpu_batch": B, "global_batch": total, "nfe": nfe, "lambda": lam,
                   "unet_batch_rows": rows, "includes": ("VAE encode (HIP kernels), " if "inversion" in name else "") +
                   "set_context (cross-attention K/V of every block; once per job), UNet + fused CFG++ step x NFE, VAE decode (HIP kernels), D2H copy",
                   "weights": "seeded synthetic, exact diffusers shapes", "build_id": _build_id(), "flops_per_image": flops_per_image,
                   "whole_path_frac_of_mfma_peak": round(value * flops_per_image / (world * PEAK_MFMA_FP16), 4)},
        "ranks": {"job_ms_min": round(min(flat) * 1e3, 2), "job_ms_max": round(max(flat) * 1e3, 2),
                  "job_ms_mean_per_rank": [round(sum(r) / len(r) * 1e3, 2) for r in per_rank],
                  "conditioning_broadcast_ms": round(broadcast_ms, 3), "tile_tuning": tuning,
                  "identity": [json.loads(x) for x in D.gather_strings(D.rank_identity(dev), dev)]},
    }
    if rank == 0 and not args.no_profile:
        result["roofline"] = roofline_block(eng, config, B, dev, PROFILE_ROUND)
    if rank == 0 and world == 1 and with_cpu_baseline and not args.no_cpu_baseline:
        log("cpu baseline (child process, bounded) ...")
        try:
            result["cpu_baseline"] = cpu_baseline(cfg_name, name, nfe, lam, img)
        except Exception as e:  # noqa: BLE001
            result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        log("cpu baseline done")
    del solver, eng
    return result


def device_for(local_rank):
    """the rank's GPU (tests/bench_mock_main.py replaces this and make_solver to run the same main() on CPU + gloo)"""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank-local device {local_rank} requested but only {torch.cuda.device_count()} GPUs are visible")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    return dev


def sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no torchrun environment): start the N ranks ourselves -
    re-exec this very command under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 - and exit with
    its status.  Rank 0 of the child job prints the JSON line on the inherited stdout.  Returns only when the process
    already is a rank (or N == 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher environment: starting {args.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
    raise SystemExit(subprocess.call(cmd, env=env))


def _build_id():
    from cfgpp_amd import _lib
    return _lib.build_id()


def main():
    args = parse()
    self_launch(args)
    from cfgpp_amd import dist as D
    rank, local_rank, world = D.init()
    if world != args.gpus:
        # never a silent 1-GPU run under an N-GPU label (or the reverse)
        raise SystemExit(f"--gpus {args.gpus} but the job has {world} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE', 'unset')})")
    dev = device_for(local_rank)
    batch, scaling = args.batch, "weak"
    if args.global_batch:
        # strong scaling: the job's total batch is fixed, every rank takes an equal share (BASELINE configs 3 - 5 as written)
        if args.batch:
            raise SystemExit("--batch (per-GPU, weak scaling) and --global-batch (total, strong scaling) exclude each other")
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} does not divide over {world} rank(s)")
        batch, scaling = args.global_batch // world, "strong"
    result = run_workload(args, args.config, rank, world, dev, args.steps, args.warmup, batch, args.nfe, scaling=scaling)
    if args.config == "sd15" and not args.no_also and not batch and not args.nfe:
        # BASELINE.json's metric names SD1.5 512^2 AND SDXL 1024^2: the default command also times a short SDXL leg
        # (configs[2]'s per-GPU share: batch 2 per GPU, 50 NFE) - 1 warm-up job (in-situ tuning) + 2 timed jobs
        import gc
        gc.collect()
        if dev.type == "cuda":
            torch.cuda.empty_cache()
        log("also: SDXL 1024x1024 leg")
        try:
            xl = run_workload(args, "sdxl", rank, world, dev, 2, 1, with_cpu_baseline=False)
            result["also"] = {"sdxl_b2": {k: xl[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling",
                                                             "dtype", "data", "config", "ranks", "roofline") if k in xl}}
        except Exception as e:  # noqa: BLE001
            if world > 1:
                raise                       # a rank that dropped out would leave the others waiting at a barrier
            result["also"] = {"sdxl_b2": {"error": f"{type(e).__name__}: {e}"}}
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        cpu_baseline_child(sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]))
    else:
        main()
