"""-m gpu, LAST in the suite (tests/conftest.py): the real SD1.5 / SDXL nets at the benchmark's own sizes against fixtures
recorded from ``oracle/`` in the build container (tests/golden/make_unet_golden.py; inputs and tolerances: tests/realsize.py).
Each case runs in its own python process: a native abort in the library fails ONE test with its stderr on show, and the
GPU box spends neither minutes nor tens of GB of host memory on a CPU oracle.  Pins the UNet calls
latent_diffusion.py:155 / latent_sdxl.py:181 and the loops latent_diffusion.py:653-674, latent_sdxl.py:730-752, 838-858.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_case(case, timeout=600):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    env = dict(os.environ, PYTHONFAULTHANDLER="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "realsize.py"), case], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("REALSIZE_RESULT ")]
    tail = (r.stdout[-1500:] + "\n--- stderr ---\n" + r.stderr[-3000:])
    assert lines, f"{case}: the child (rc {r.returncode}) printed no result\n{tail}"
    res = json.loads(lines[-1][len("REALSIZE_RESULT "):])
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"parity_realsize_{os.getppid()}_{os.getpid()}.jsonl"), "a") as f:
            f.write(json.dumps(res) + "\n")
    except OSError:
        pass
    assert r.returncode == 0 and res["ok"], f"{case}: {res}\n{tail}"
    return res


@pytest.mark.parametrize("case", ["sd15_fwd_r2", "sdxl_fwd_r2_32"])
def test_real_nets_at_small_row_counts_vs_oracle_fixture(case):
    """SD1.5 2 rows @ 64 x 64 and SDXL 2 rows @ 32 x 32 latents at t = 981 and t = 1 (other plans, other K-split / tile rules than the
    bench sizes)"""
    res = run_case(case)
    assert set(res) >= {"t981", "t1"}


def test_real_sd15_forward_16_rows_vs_oracle_fixture():
    """C2's forward (16 rows @ 64 x 64, autotuned tiles, K-split 8 x 8 level, d = 40 attention over 4096 tokens)"""
    res = run_case("sd15_fwd")
    assert res["repeat_bit_identical"]


def test_real_sdxl_forward_at_every_bench_plan_size_vs_oracle_fixture():
    """SDXL @ 128 x 128 at 4 (C3), 2 (C5) and 16 (C4) rows: every output row of every plan against its own oracle row"""
    res = run_case("sdxl_fwd")
    assert set(res) >= {"rows2", "rows4", "rows16"}


def test_real_sd15_batch8_chain_4_steps_vs_oracle_fixture():
    run_case("sd15_chain")


def test_real_sd15_chain_as_graph_replay_vs_oracle_fixture():
    """the whole-loop hipGraph replay on the real SD1.5 net at batch 8: the fixture's tolerance, and bit-identical to the eager loop"""
    res = run_case("sd15_chain_graph")
    assert res["graph_equals_eager"]


def test_real_sdxl_chains_vs_oracle_fixture():
    """C3 (2 NFE ddim_cfg++, batch 2) and C4 (1 NFE ddim_cfg++_lightning: positive rows only)"""
    res = run_case("sdxl_chain")
    assert res["ddim_cfg++_lightning"]["rows_seen"] == 1 and res["ddim_cfg++"]["rows_seen"] == 4
