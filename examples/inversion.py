"""Command-line twin of the reference's examples/inversion.py (same flags and defaults) on the MI355X path:
encode an image, DDIM-invert it under the source prompt and reconstruct it (``ddim_inversion`` /
``ddim_inversion_cfg++`` / ``ddim_edit*``).

    python examples/inversion.py --img_path cat.jpg --prompt "a photo of a cat" --method ddim_inversion_cfg++ \
        --cfg_guidance 0.6 --NFE 10 [--model sd15|sdxl] [--unet_weights ... --vae_weights ...]

Additive flags as in text_to_img.py.  The VAE posterior noise is drawn from the CPU generator seeded by
``--seed`` (the reference samples it on the device, which is not reproducible across devices).
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_img(img_path, size: int = 512) -> torch.Tensor:
    """RGB image file -> [1,3,size,size] float in [-1, 1] (what ``solver.encode`` expects)."""
    from PIL import Image
    arr = np.asarray(Image.open(img_path).convert("RGB").resize((size, size)), dtype=np.float32)
    return (torch.from_numpy(arr).permute(2, 0, 1) / 127.5 - 1.0).unsqueeze(0)


# (flag, type, default) - the reference CLI's flags with its defaults, then ours
REFERENCE_FLAGS = (
    ("workdir", Path, Path("examples/workdir/inversion")), ("img_path", Path, Path("examples/assets/afhq_1.jpg")),
    ("img_size", int, 512), ("device", str, "cuda"), ("null_prompt", str, ""), ("prompt", str, ""),
    ("cfg_guidance", float, 7.5), ("method", str, "ddim_inversion_cfg++"), ("NFE", int, 10), ("seed", int, 42),
)
EXTRA_FLAGS = (("unet_weights", str, "synthetic"), ("vae_weights", str, None), ("model_dir", str, None))


def main(argv=None, solver_kwargs=None) -> None:
    """``solver_kwargs`` lets tests inject ``engine=`` / ``vae=`` (CPU mock); the CLI never passes it."""
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for flag, kind, default in REFERENCE_FLAGS + EXTRA_FLAGS:
        ap.add_argument(f"--{flag}", type=kind, default=default)
    ap.add_argument("--model", default="sd15", choices=("sd15", "sd20", "sdxl"))
    args = ap.parse_args(argv)

    from cfgpp_amd.callback_util import save_image
    (args.workdir / "result").mkdir(parents=True, exist_ok=True)
    torch.manual_seed(args.seed)
    xl = args.model == "sdxl"
    size = args.img_size if not xl or args.img_size != 512 else 1024
    img = load_img(args.img_path, size)
    kw = dict(solver_config=types.SimpleNamespace(num_sampling=args.NFE), device=args.device, max_batch=1,
              unet_weights=args.unet_weights, latent_hw=(size // 8, size // 8))
    if args.vae_weights:
        kw["vae_weights"] = args.vae_weights
    if args.model_dir:        # a local diffusers-layout checkpoint: UNet / VAE weights, CLIP tower(s) + BPE tokenizer(s)
        from cfgpp_amd.checkpoint import solver_kwargs_from_dir
        found, missing = solver_kwargs_from_dir(args.model_dir, args.model in ("sdxl", "sdxl_lightning"), args.device)
        if missing:
            print(f"--model_dir {args.model_dir}: no {', '.join(missing)} there - synthetic stand-in(s) used")
        for k, v in found.items():
            if k.endswith("_weights") and getattr(args, k, None) not in (None, "synthetic"):
                continue                                   # an explicit --unet_weights / --vae_weights wins
            kw[k] = v
    kw.update(solver_kwargs or {})
    if xl:
        from cfgpp_amd.latent_sdxl import get_solver
        solver = get_solver(args.method, **kw)
        result = solver.sample(prompt1=[args.null_prompt, args.prompt, args.prompt], prompt2=[args.null_prompt, args.prompt, args.prompt],
                               src_img=img, cfg_guidance=args.cfg_guidance, target_size=(size, size), callback_fn=None)
    else:
        from cfgpp_amd.latent_diffusion import get_solver
        solver = get_solver(args.method, **kw)
        result = solver.sample(prompt=[args.null_prompt, args.prompt], src_img=img, cfg_guidance=args.cfg_guidance, callback_fn=None)
    save_image(result, args.workdir / "result" / "reconstruct.png", normalize=True)
    print(f"saved {args.workdir / 'result' / 'reconstruct.png'}")


if __name__ == "__main__":
    main()
