"""Command-line twin of the reference's examples/text_to_img.py (same flags and defaults) on the MI355X path.

    python examples/text_to_img.py --prompt "a corgi" --method ddim_cfg++ --cfg_guidance 0.6 --NFE 50 \
        [--model sd15|sdxl|sdxl_lightning] [--unet_weights unet.safetensors --vae_weights vae.safetensors] \
        [--batch 8] [--draw]

Differences from the reference, all additive: ``--unet_weights / --vae_weights`` (diffusers-layout safetensors;
default = seeded synthetic weights, because no checkpoint exists offline), ``--batch`` (B chains with seeds
seed, seed+1, ... in one UNet batch of 2B rows) and ``--draw`` (the reference keeps its ComposeCallback
commented out).  Text goes through ``solver.text_encoder`` (synthetic embeddings unless a CLIP callable is
plugged in, cfgpp_amd/conditioning.py).  Reference behaviour kept: CPU-generator initial latent from
``--seed``, default null prompt, 1024x1024 target for SDXL, result saved (min-max normalised, as torchvision's
``save_image(normalize=True)`` does there) to <workdir>/result/generated.png.
"""
from __future__ import annotations

import argparse
import os
import sys
import types
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# (flag, type, default) - the reference CLI's flags with its defaults, then ours
REFERENCE_FLAGS = (
    ("workdir", Path, Path("examples/workdir/t2i")), ("device", str, "cuda"),
    ("null_prompt", str, "low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,"), ("prompt", str, ""),
    ("cfg_guidance", float, 7.5), ("method", str, "ddim"), ("NFE", int, 50), ("seed", int, 42),
)
EXTRA_FLAGS = (("unet_weights", str, "synthetic"), ("vae_weights", str, None), ("model_dir", str, None), ("batch", int, 1))


def main(argv=None, solver_kwargs=None) -> None:
    """``solver_kwargs`` lets tests inject ``engine=`` / ``vae=`` (CPU mock); the CLI never passes it."""
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    for flag, kind, default in REFERENCE_FLAGS + EXTRA_FLAGS:
        ap.add_argument(f"--{flag}", type=kind, default=default)
    ap.add_argument("--model", default="sd15", choices=("sd15", "sd20", "sdxl", "sdxl_lightning"))
    ap.add_argument("--draw", action="store_true", help="save z0t / zt decodes every step (draw_tweedie + draw_noisy)")
    args = ap.parse_args(argv)

    from cfgpp_amd.callback_util import ComposeCallback, save_image
    (args.workdir / "result").mkdir(parents=True, exist_ok=True)
    torch.manual_seed(args.seed)
    cfg = types.SimpleNamespace(num_sampling=args.NFE)
    callback = ComposeCallback(workdir=args.workdir, frequency=1, callbacks=["draw_noisy", "draw_tweedie"]) if args.draw else None
    kw = dict(solver_config=cfg, device=args.device, max_batch=args.batch, unet_weights=args.unet_weights)
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
    prompts = [args.prompt] * args.batch if args.batch > 1 else args.prompt
    seeds = None if args.batch == 1 else [args.seed + i for i in range(args.batch)]   # B = 1: global CPU RNG, like the reference

    if args.model in ("sdxl", "sdxl_lightning"):
        from cfgpp_amd.latent_sdxl import get_solver
        solver = get_solver(args.method, **kw)
        result = solver.sample(prompt1=[args.null_prompt, prompts], prompt2=[args.null_prompt, prompts],
                               cfg_guidance=args.cfg_guidance, target_size=(1024, 1024), callback_fn=callback, seeds=seeds)
    else:                                   # "sd20" is accepted and runs SD1.5, like the reference (quirk Q8)
        from cfgpp_amd.latent_diffusion import get_solver
        solver = get_solver(args.method, **kw)
        result = solver.sample(prompt=[args.null_prompt, prompts], cfg_guidance=args.cfg_guidance, callback_fn=callback, seeds=seeds)

    for i in range(result.shape[0]):
        name = "generated.png" if result.shape[0] == 1 else f"generated_{i}.png"
        save_image(result[i:i + 1], args.workdir / "result" / name, normalize=True)
    print(f"saved {result.shape[0]} image(s) to {args.workdir / 'result'}")


if __name__ == "__main__":
    main()
