import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # as cfgpp_amd/__init__.py (must precede the first HIP call)
os.environ.setdefault("CFGPP_TUNE_CACHE", "0")           # hermetic tests: no tile pins from / to ~/.cache (test_gpu_unet.py tests the cache itself)

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "sampler_golden.json")) as f:
        meta = json.load(f)
    return g, meta


@pytest.fixture(scope="session")
def golden_h16():
    """inversion / edit trajectories recorded from the reference with an fp16 VAE (fp16 latents end to end)"""
    import json

    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden_h16.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "sampler_golden_h16.json")) as f:
        meta = json.load(f)
    return g, meta


@pytest.fixture(scope="session")
def golden_r3():
    """round-3 additions recorded from the reference: SDXL euler / *_lightning trajectories, the 'npi' initialisation"""
    import json

    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden_r3.npz"))
    with open(os.path.join(ROOT, "tests", "golden", "sampler_golden_r3.json")) as f:
        meta = json.load(f)
    return g, meta


# ---- order of the -m gpu suite: cheap -> expensive -------------------------------------------------------------------
# The driver runs `pytest tests/ -x -q -m gpu`: with -x a failure (or a native abort) hides every test behind it, so the
# single-kernel parity tests go first and the real-size nets (minutes each, tens of GB of host memory) go last.
_GPU_FILE_ORDER = ["test_gpu_kernels.py", "test_gpu_step.py", "test_gpu_torch_semantics.py", "test_gpu_graph.py",
                   "test_gpu_unet.py", "test_gpu_vae.py", "test_gpu_text.py", "test_gpu_weights.py", "test_gpu_examples.py",
                   "test_gpu_configs.py", "test_gpu_realsize.py"]


def pytest_collection_modifyitems(config, items):
    if os.environ.get("CFGPP_TEST_ORDER") == "alpha":       # diagnostics: pytest's own order
        return

    def key(it_idx):
        idx, it = it_idx
        fname = os.path.basename(str(it.fspath))
        if it.get_closest_marker("gpu") is None or fname not in _GPU_FILE_ORDER:
            return (0, 0, idx)                                 # CPU tests and unknown files keep their place, first
        return (1, _GPU_FILE_ORDER.index(fname), idx)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]
