"""Record the real-size oracle outputs ONCE, in the build container (CPU), as fixtures for tests/test_gpu_realsize.py.

    python tests/golden/make_unet_golden.py [case ...]        # cases: sd15_fwd sdxl_fwd sd15_chain sdxl_chain (default: all)

Imports only this repo's own ``oracle/`` restatement (fp32 torch on CPU) through ``tests/realsize.py``, which also defines
the seeded inputs; nothing of /root/reference is involved (the reference delegates the UNet to diffusers, which is not
installable here - see oracle/unet_ref.py's header: these fixtures pin the HIP path to the restatement, the restatement
itself stays "parity unpinned").  Costs ~2 / ~4 / ~7 / ~9 minutes per case on 8 cores and up to ~25 GB of host memory (SDXL).
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import realsize as RS  # noqa: E402

import numpy as np  # noqa: E402


def main():
    cases = sys.argv[1:] or list(RS.ORACLE)
    meta_path = os.path.join(HERE, "realsize_golden.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    for case in cases:
        t0 = time.time()
        out = RS.ORACLE[case]()
        np.savez(RS.fixture_path(case), **out)
        meta[case] = dict(arrays={k: [list(v.shape), str(v.dtype)] for k, v in out.items()}, oracle_cpu_s=round(time.time() - t0, 1),
                          torch=__import__("torch").__version__, threads=__import__("torch").get_num_threads())
        print(case, meta[case], flush=True)
        with open(meta_path, "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
