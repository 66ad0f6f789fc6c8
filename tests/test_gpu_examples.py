"""-m gpu: the example command lines (twins of the reference's examples/text_to_img.py, inversion.py, text_to_mscoco.py)
end to end on the HIP engines - real-size SD1.5 UNet + VAE with seeded synthetic weights, a few NFE: the flags of the
reference's README commands, PNGs of the reference's sizes, finite non-constant pixels.  (The same CLIs run against a CPU
mock engine in tests/test_examples_cpu.py; the arithmetic is pinned elsewhere - this is the plumbing on the device.)"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


@pytest.fixture(scope="module", autouse=True)
def _needs_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _check_png(path, size):
    from PIL import Image
    im = Image.open(path)
    assert im.size == size and im.mode == "RGB"
    a = np.asarray(im).astype(np.float32)
    assert a.std() > 1.0, "constant image"
    return a


def test_text_to_img_cli_on_the_hip_engines(tmp_path):
    import text_to_img
    # the reference's README command (python -m examples.text_to_img --prompt ... --method ddim_cfg++ --cfg_guidance 0.6), fewer NFE
    text_to_img.main(["--prompt", "a portrait of a dog", "--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "4",
                      "--batch", "2", "--workdir", str(tmp_path)])
    a = _check_png(tmp_path / "result" / "generated_0.png", (512, 512))
    b = _check_png(tmp_path / "result" / "generated_1.png", (512, 512))
    assert not np.array_equal(a, b)                      # seeds 42 and 43: different chains of one UNet batch
    torch.cuda.synchronize()


def test_inversion_cli_on_the_hip_engines(tmp_path):
    from PIL import Image
    import inversion
    rng = np.random.default_rng(3)
    src = tmp_path / "src.png"
    Image.fromarray(rng.integers(0, 255, (96, 96, 3), dtype=np.uint8)).resize((512, 512)).save(src)
    inversion.main(["--img_path", str(src), "--prompt", "a photo", "--method", "ddim_inversion_cfg++", "--cfg_guidance", "0.6",
                    "--NFE", "4", "--workdir", str(tmp_path)])
    pngs = sorted((tmp_path / "result").glob("*.png"))
    assert pngs, "no result written"
    for p in pngs:
        _check_png(p, (512, 512))
    torch.cuda.synchronize()


def test_mscoco_cli_on_the_hip_engines(tmp_path):
    import text_to_mscoco
    caps = tmp_path / "caps.txt"
    caps.write_text("a cat\n\na dog on a sofa\na red bus\n")
    out = tmp_path / "coco"
    n = text_to_mscoco.main(["--prompt_dir", str(caps), "--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "3",
                             "--batch", "2", "--no_draw", "--workdir", str(out)])
    assert n == 3 and sorted(p.name for p in out.glob("*.png")) == ["00000.png", "00001.png", "00002.png"]
    for p in out.glob("*.png"):
        _check_png(p, (512, 512))
    torch.cuda.synchronize()
