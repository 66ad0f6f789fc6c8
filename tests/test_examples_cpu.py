"""The example CLIs (twins of the reference's examples/text_to_img.py / inversion.py) end to end on the CPU with
an injected mock engine and stub VAE: flag parsing, solver plumbing, PNG output."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

from mock_engine import MockEngine, StubVAE          # noqa: E402


def _unet(z, t, ehs, te, ti):
    return (0.1 * z + 0.01 * ehs.float().mean(dim=(1, 2)).view(-1, 1, 1, 1)).half()


def test_text_to_img_cli_writes_pngs(tmp_path):
    from PIL import Image
    import text_to_img
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215), latent_hw=(8, 8))
    text_to_img.main(["--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "3", "--prompt", "a cat", "--device", "cpu",
                      "--batch", "2", "--draw", "--workdir", str(tmp_path)], solver_kwargs=kw)
    for i in range(2):
        im = Image.open(tmp_path / "result" / f"generated_{i}.png")
        assert im.size == (64, 64) and im.mode == "RGB"
    assert len(list((tmp_path / "record").rglob("*.png"))) > 0          # draw_* callbacks fired


def test_inversion_cli_reconstructs(tmp_path):
    from PIL import Image
    import inversion
    src = tmp_path / "src.png"
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(src)
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215))
    inversion.main(["--img_path", str(src), "--img_size", "64", "--method", "ddim_inversion_cfg++", "--cfg_guidance", "0.6",
                    "--NFE", "4", "--prompt", "noise", "--device", "cpu", "--workdir", str(tmp_path)], solver_kwargs=kw)
    im = Image.open(tmp_path / "result" / "reconstruct.png")
    assert im.size == (64, 64)
    assert torch.isfinite(torch.from_numpy(np.asarray(im, dtype=np.float32))).all()


def test_solver_with_clip_text_tower():
    """the opt-in CLIP tower plugs into the SD1.5 solver through the same ``text_encoder=`` seam as the synthetic one"""
    import types
    from cfgpp_amd.conditioning import ClipTextTower
    from cfgpp_amd.latent_diffusion import get_solver
    eng = MockEngine(_unet, (8, 8))
    s = get_solver("ddim_cfg++", solver_config=types.SimpleNamespace(num_sampling=2), device="cpu", engine=eng,
                   text_encoder=ClipTextTower.clip_l(layers=1), latent_hw=(8, 8), vae=StubVAE(0.18215))
    img = s.sample(prompt=["", "a cat"], cfg_guidance=0.6)
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    assert eng.ehs.shape == (2, 77, 768) and eng.ehs.dtype == torch.float16
