"""A diffusers-layout checkpoint directory -> the keyword arguments of ``get_solver``.

The reference builds everything with ``StableDiffusionPipeline.from_pretrained(model_key)`` /
``StableDiffusionXLPipeline.from_pretrained(model_key)`` (latent_diffusion.py:56-66, latent_sdxl.py:40-54).  With a local
copy of such a checkpoint (no hub access here) the same components map onto this package as

    <dir>/unet/diffusion_pytorch_model.safetensors          -> unet_weights=   (HIP UNet engine)
    <dir>/vae/diffusion_pytorch_model.safetensors           -> vae_weights=    (HIP VAE engine)
    <dir>/text_encoder/{config.json, model.safetensors} + <dir>/tokenizer/{vocab.json, merges.txt}        -> text_encoder=
    <dir>/text_encoder_2/... + <dir>/tokenizer_2/...        (SDXL: second tower, projected pooled output, "!" padding)

Missing pieces are left to the solver's defaults (synthetic weights / synthetic text encoder) and reported in ``missing``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch


def _first(*paths):
    for p in paths:
        if os.path.exists(p):
            return p
    return None


def solver_kwargs_from_dir(model_dir, sdxl: bool, device="cuda") -> Tuple[Dict, List[str]]:
    """(kwargs for ``get_solver``, names of the components not found)."""
    from .conditioning import ClipTextTower
    d = str(model_dir)
    kw, missing = {}, []
    unet = _first(os.path.join(d, "unet", "diffusion_pytorch_model.safetensors"), os.path.join(d, "unet", "diffusion_pytorch_model.fp16.safetensors"))
    vae = _first(os.path.join(d, "vae", "diffusion_pytorch_model.safetensors"), os.path.join(d, "vae", "diffusion_pytorch_model.fp16.safetensors"))
    if unet:
        kw["unet_weights"] = unet
    else:
        missing.append("unet")
    if vae:
        kw["vae_weights"] = vae
    else:
        missing.append("vae")
    on_gpu = torch.device(device).type == "cuda"
    dtype = torch.float16 if on_gpu else torch.float32          # the reference runs the text encoders in the pipeline's fp16

    def tower(enc, tok, **args):
        e, t = os.path.join(d, enc), os.path.join(d, tok)
        ok = all(os.path.exists(os.path.join(e, f)) for f in ("config.json", "model.safetensors")) and \
            all(os.path.exists(os.path.join(t, f)) for f in ("vocab.json", "merges.txt"))
        if not ok:
            missing.append(enc)
            return None
        return ClipTextTower.from_dir(e, t, device=device, dtype=dtype, **args)

    if sdxl:
        t1 = tower("text_encoder", "tokenizer", penultimate=True, with_projection=False)
        t2 = tower("text_encoder_2", "tokenizer_2", penultimate=True, with_projection=True, pad_token="!")
        if t1 is not None and t2 is not None:
            kw["text_encoder"] = (t1, t2)
    else:
        t1 = tower("text_encoder", "tokenizer", penultimate=False, with_projection=False)
        if t1 is not None:
            kw["text_encoder"] = t1
    return kw, missing
