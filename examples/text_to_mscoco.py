"""Command-line twin of the reference's examples/text_to_mscoco.py (the 10k-caption MS-COCO sweep behind its FID table)
on the MI355X path: one image per caption of ``--prompt_dir``, saved as <workdir>/00000.png, 00001.png, ...

    python examples/text_to_mscoco.py --prompt_dir captions.txt --model sdxl --method ddim_cfg++ --cfg_guidance 0.6 \
        [--batch 8] [--limit 100] [--unet_weights unet.safetensors --vae_weights vae.safetensors]

Reference behaviour kept: same flags and defaults (``--null_prompt ""``, ``--method ddim``, ``--NFE 50``, ``--seed 42``,
draw_noisy / draw_tweedie callbacks on), blank caption lines skipped, at most the first 10 000 captions, file names are the
zero-padded caption index, 1024x1024 target for SDXL, ``save_image(normalize=True)``.  The reference only has an SDXL
branch (``--model sd15`` parses and then writes nothing); here sd15 / sd20 run the SD1.5 solver as text_to_img.py does.

Additive: ``--batch B`` runs B captions per UNet batch of 2B rows (the null prompt is embedded once and broadcast - this
is the batched hot path bench.py times); ``--limit``; ``--unet_weights / --vae_weights``; ``--no_draw``.
Seeding: the reference seeds once and lets the global CPU RNG run on from caption to caption; with ``--batch 1`` this
driver does exactly that.  With B > 1 caption i gets the explicit seed ``seed + i`` (independent chains in one batch).
Under ``torchrun`` every rank takes the contiguous shard of captions cfgpp_amd.dist hands it (no collective in the loop).
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

REFERENCE_FLAGS = (
    ("workdir", Path, Path("examples/workdir/mscoco")), ("prompt_dir", Path, Path("examples/assets/coco_v2.txt")),
    ("device", str, "cuda"), ("null_prompt", str, ""), ("prompt", str, ""), ("cfg_guidance", float, 7.5),
    ("method", str, "ddim"), ("NFE", int, 50), ("seed", int, 42),
)
EXTRA_FLAGS = (("unet_weights", str, "synthetic"), ("vae_weights", str, None), ("model_dir", str, None), ("batch", int, 1), ("limit", int, 10000))


def read_captions(path: Path, limit: int = 10000) -> list:
    """non-blank stripped lines, first ``limit`` (the reference keeps 10 000: its MS-COCO validation subset)"""
    with open(path, "r") as f:
        captions = [line.strip() for line in f if line.strip()]
    return captions[:limit]


def main(argv=None, solver_kwargs=None) -> int:
    """``solver_kwargs`` lets tests inject ``engine=`` / ``vae=`` (CPU mock); the CLI never passes it.  Returns the
    number of images this process wrote."""
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for flag, kind, default in REFERENCE_FLAGS + EXTRA_FLAGS:
        ap.add_argument(f"--{flag}", type=kind, default=default)
    ap.add_argument("--model", default="sd15", choices=("sd15", "sd20", "sdxl", "sdxl_lightning"))
    ap.add_argument("--no_draw", action="store_true", help="skip the per-step draw_noisy / draw_tweedie decodes")
    args = ap.parse_args(argv)

    from cfgpp_amd import dist
    from cfgpp_amd.callback_util import ComposeCallback, save_image
    args.workdir.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(args.seed)
    captions = read_captions(args.prompt_dir, min(args.limit, 10000))
    rank, _, world = dist.env_world()
    lo, hi = dist.shard_range(len(captions), rank, world)               # contiguous shard per rank, nothing to exchange
    B = max(1, args.batch)

    cfg = types.SimpleNamespace(num_sampling=args.NFE)
    callback = None if args.no_draw else ComposeCallback(workdir=args.workdir, frequency=1, callbacks=["draw_noisy", "draw_tweedie"])
    kw = dict(solver_config=cfg, device=args.device, max_batch=B, unet_weights=args.unet_weights)
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
    xl = args.model in ("sdxl", "sdxl_lightning")
    if xl:
        from cfgpp_amd.latent_sdxl import get_solver
    else:
        from cfgpp_amd.latent_diffusion import get_solver
    solver = get_solver(args.method, **kw)

    written = 0
    for start in range(lo, hi, B):
        idx = list(range(start, min(start + B, hi)))
        texts = [captions[i] for i in idx]
        for i, text in zip(idx, texts):
            print(f"Processing {i + 1}/{len(captions)}: {text}")
        prompts = texts[0] if B == 1 else texts
        seeds = None if B == 1 else [args.seed + i for i in idx]        # B = 1: the global CPU RNG runs on, like the reference
        if xl:
            result = solver.sample(prompt1=[args.null_prompt, prompts], prompt2=[args.null_prompt, prompts],
                                   cfg_guidance=args.cfg_guidance, target_size=(1024, 1024), callback_fn=callback, seeds=seeds)
        else:
            result = solver.sample(prompt=[args.null_prompt, prompts], cfg_guidance=args.cfg_guidance, callback_fn=callback, seeds=seeds)
        for k, i in enumerate(idx):
            save_image(result[k:k + 1], args.workdir / f"{str(i).zfill(5)}.png", normalize=True)
            written += 1
    print(f"rank {rank}/{world}: wrote {written} image(s) to {args.workdir}")
    return written


if __name__ == "__main__":
    main()
