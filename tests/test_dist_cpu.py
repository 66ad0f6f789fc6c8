"""N > 1 path on CPU: world_size 2, gloo.  Shards, the single conditioning broadcast, gather."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cfgpp_amd import dist as D
    r, lr, w = D.init("gloo")
    total = 5
    shapes = [((1, 77, 64), torch.float16), ((total, 77, 64), torch.float16), ((total, 6), torch.float32)]
    payload = [None, None, None]
    if r == 0:
        g = torch.Generator().manual_seed(0)
        payload = [torch.randn(s, generator=g).to(dt) for s, dt in shapes]
    got = D.broadcast_conditioning(payload, shapes, torch.device("cpu"))
    lo, hi = D.shard_range(total, r, w)
    local = got[1][lo:hi].float().sum(dim=(1, 2), keepdim=False).reshape(-1, 1)      # a per-chain "result"
    counts = [D.shard_range(total, i, w)[1] - D.shard_range(total, i, w)[0] for i in range(w)]
    gathered = D.gather_rows(local, counts)
    mx = D.max_over_ranks(float(r + 1), torch.device("cpu"))
    hints = D.broadcast_ints([5, 0, 14, 69] if r == 0 else None, torch.device("cpu"))      # rank 0's pinned tile configs
    times = D.gather_floats([1.0 + r, 2.0 + r], torch.device("cpu"))
    D.barrier()
    q.put((r, [float(x.float().sum()) for x in got], (lo, hi), None if gathered is None else gathered.reshape(-1).tolist(), mx, hints, times))


def test_world2_broadcast_shard_gather():
    world, port = 2, 29533
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert res[0][1] == res[1][1]                       # every rank holds the same conditioning
    assert res[0][2] == (0, 3) and res[1][2] == (3, 5)  # contiguous shards, sizes differ by <= 1
    g = torch.Generator().manual_seed(0)
    ref = [torch.randn(s, generator=g).to(dt) for s, dt in [((1, 77, 64), torch.float16), ((5, 77, 64), torch.float16), ((5, 6), torch.float32)]]
    want = ref[1].float().sum(dim=(1, 2)).tolist()
    assert res[0][3] is not None and res[1][3] is None
    assert all(abs(a - b) < 1e-3 for a, b in zip(res[0][3], want))
    assert res[0][4] == res[1][4] == 2.0
    assert res[0][5] == res[1][5] == [5, 0, 14, 69]
    assert res[0][6] == res[1][6] == [[1.0, 2.0], [2.0, 3.0]]


def test_shard_range_covers_everything():
    from cfgpp_amd.dist import shard_range
    for n in (1, 7, 16, 64):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- bench.py's own multi-GPU flow (broadcast -> shard -> solver.sample -> gather) on CPU, world 2, gloo ----
def _bench_flow(kind, rank, world, B):
    """what bench.main() does per rank, with a CPU mock engine in place of the HIP engine"""
    import types
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    from _stub_env import pointwise_eps
    from mock_engine import MockEngine, StubVAE
    from cfgpp_amd.unet_config import TINY_SD, TINY_XL

    def unet(z, t, ehs, te, ti):
        return pointwise_eps(z, torch.as_tensor(float(t)).reshape(1), ehs, te, ti)
    sc = types.SimpleNamespace(num_sampling=3)
    if kind == "sd":
        from cfgpp_amd.latent_diffusion import get_solver
        cfg, name = TINY_SD, "ddim_cfg++"
    else:
        from cfgpp_amd.latent_sdxl import get_solver
        cfg, name = TINY_XL, "ddim_cfg++"
    solver = get_solver(name, solver_config=sc, device="cpu", unet_config=cfg, max_batch=B, latent_hw=(8, 8),
                        engine=MockEngine(unet, (8, 8)), vae=StubVAE(cfg.vae_scale))
    one_job, total = bench.prepare_job(solver, cfg, kind, name, B, 64, 0.6, rank, world, torch.device("cpu"))
    out = one_job(return_latents=True)
    z = out[0] if kind == "sd" else out
    img = one_job()                              # the decode + D2H leg as well
    assert img.shape[0] == B and bool(torch.isfinite(img).all())
    return z.float(), total


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cfgpp_amd import dist as D
    r, lr, w = D.init("gloo")
    res = {}
    for kind in ("sd", "xl"):
        z, total = _bench_flow(kind, r, w, 2)
        allz = D.gather_rows(z, [2] * w)
        dt = D.max_over_ranks(0.5 + r, torch.device("cpu"))
        res[kind] = (None if allz is None else allz.numpy(), total, dt)
    D.barrier()
    q.put((r, res))


def test_world2_bench_flow_equals_single_process():
    """`bench.py --gpus 2` plumbing: every rank samples its own prompt / seed shard from ONE broadcast conditioning
    block; the gathered latents equal the single-process run of the same global batch, chain by chain."""
    world, port = 2, 29547
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for kind in ("sd", "xl"):
        got, total, dt = res[0][kind]
        assert res[1][kind][0] is None and total == 4 and dt == res[1][kind][2] == 1.5
        ref, _ = _bench_flow(kind, 0, 1, 4)      # world 1, the whole global batch in one process
        assert torch.equal(torch.from_numpy(got), ref), kind


# ---- `python bench.py --gpus N` started without a launcher must start N ranks itself ----
def _run_mock_bench(argv, env_extra=None, timeout=600):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    env["CFGPP_BENCH_VERBOSE"] = "0"
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_mock_main.py")] + argv, capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_bench_gpus2_self_launches_two_ranks():
    """the driver's command line, no torchrun around it: bench.main() re-executes itself under torch.distributed.run with two
    ranks (gloo here, RCCL on the GPU box) and rank 0 prints ONE JSON line that says n_gpus 2 and carries both ranks' job times"""
    import json
    p = _run_mock_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "mock", "--no-profile", "--no-cpu-baseline", "--no-also"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 2 * r["config"]["per_gpu_batch"]
    assert len(r["ranks"]["job_ms_mean_per_rank"]) == 2
    assert "3 pins broadcast" in r["ranks"]["tile_tuning"]
    ident = r["ranks"]["identity"]                 # one record per rank: a SCALE line can show that N ranks really ran
    assert [i["rank"] for i in ident] == [0, 1] and all(i["world"] == 2 and i["backend"] == "gloo" for i in ident)
    assert len({i["pid"] for i in ident}) == 2


def test_bench_global_batch_is_strong_scaling():
    """--global-batch G (BASELINE configs 3 - 5 are GLOBAL batches): G / N chains per rank, labelled "strong"; a G that does not
    divide over the ranks, or combined with the per-GPU --batch, is refused"""
    import json
    common = ["--steps", "1", "--warmup", "0", "--config", "mock", "--no-profile", "--no-cpu-baseline", "--no-also"]
    p = _run_mock_bench(["--gpus", "2", "--global-batch", "4"] + common)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert r["config"]["global_batch"] == 4 and r["config"]["per_gpu_batch"] == 2
    p1 = _run_mock_bench(["--gpus", "1", "--global-batch", "4"] + common)
    assert p1.returncode == 0, p1.stderr[-2000:]
    r1 = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][0])
    assert r1["scaling"] == "strong" and r1["config"]["global_batch"] == 4 and r1["config"]["per_gpu_batch"] == 4
    bad = _run_mock_bench(["--gpus", "2", "--global-batch", "3"] + common)
    assert bad.returncode != 0 and "does not divide" in (bad.stderr + bad.stdout)
    both = _run_mock_bench(["--gpus", "1", "--global-batch", "4", "--batch", "2"] + common)
    assert both.returncode != 0 and "exclude each other" in (both.stderr + both.stdout)


def test_bench_world_size_mismatch_is_an_error():
    """--gpus 2 inside a 1-rank launcher environment (or the reverse) is refused, never a silently mislabelled run"""
    p = _run_mock_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "mock", "--no-profile", "--no-cpu-baseline", "--no-also"],
                        env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 2 but the job has 1 rank" in (p.stderr + p.stdout)
