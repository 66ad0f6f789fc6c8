"""Test driver: ``bench.main()`` with a CPU mock engine in place of the HIP engine, so the launcher logic of
``python bench.py --gpus N`` (self-launch of N ranks, the world-size check, the per-rank fields of the JSON line) can run
on a box without GPUs over gloo.  bench.py itself contains no mock path; this file replaces two of its functions.

    python tests/bench_mock_main.py --gpus 2 --steps 1 --warmup 0 --config mock --no-profile --no-cpu-baseline --no-also
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def make_solver(kind, name, cfg_name, nfe, batch, device):
    from _stub_env import pointwise_eps
    from mock_engine import MockEngine, StubVAE
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD

    def unet(z, t, ehs, te, ti):
        return pointwise_eps(z, torch.as_tensor(float(t)).reshape(1), ehs, te, ti)
    eng = MockEngine(unet, (8, 8))
    eng.device_bytes = lambda: 0.0
    eng.export_tuning = lambda: [5, 0, 14]
    eng.import_tuning = lambda hints, batch: None
    eng.flops_per_forward = lambda rows: 1.0e6 * rows
    sc = types.SimpleNamespace(num_sampling=nfe)
    solver = get_solver(name, solver_config=sc, device="cpu", unet_config=TINY_SD, max_batch=batch, latent_hw=(8, 8),
                        engine=eng, vae=StubVAE(TINY_SD.vae_scale))
    return solver, TINY_SD


bench.WORKLOADS["mock"] = ("sd", "ddim_cfg++", "tiny", 3, 0.6, 2, 64, "CPU mock engine (launcher test only)")
bench.make_solver = make_solver
bench.device_for = lambda local_rank: torch.device("cpu")

if __name__ == "__main__":
    bench.main()
