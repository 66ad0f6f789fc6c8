"""A diffusers-layout checkpoint directory -> the keyword arguments of ``get_solver``.

The reference builds everything with ``StableDiffusionPipeline.from_pretrained(model_key)`` /
``StableDiffusionXLPipeline.from_pretrained(model_key)`` (latent_diffusion.py:56-66, latent_sdxl.py:40-54).  With a local
copy of such a checkpoint (no hub access here) the same components map onto this package as

    <dir>/unet/diffusion_pytorch_model.safetensors          -> unet_weights=   (HIP UNet engine)
    <dir>/vae/diffusion_pytorch_model.safetensors           -> vae_weights=    (HIP VAE engine; SD1.5 only)
    <dir>/vae_fp16_fix/diffusion_pytorch_model.safetensors  -> vae_weights=    (SDXL: the reference REPLACES the pipeline's VAE with
                                                               madebyollin/sdxl-vae-fp16-fix, latent_sdxl.py:44,396 - the stock
                                                               SDXL VAE overflows in fp16, which is what the HIP VAE computes in;
                                                               <dir>/vae is never used for SDXL unless vae_dir= says so)
    <dir>/text_encoder/{config.json, model.safetensors} + <dir>/tokenizer/{vocab.json, merges.txt}        -> text_encoder=
    <dir>/text_encoder_2/... + <dir>/tokenizer_2/...        (SDXL: second tower, projected pooled output, "!" padding)

On a GPU device the text towers are ``text.HipClipTextTower`` - the CLIP transformer on the engine's own kernels
(csrc/text.hip), pinned against ``transformers`` in tests/test_gpu_text.py; the torch-ops ``ClipTextTower`` is what a CPU
device (tests only) gets.

Missing pieces are left to the solver's defaults (synthetic weights / synthetic text encoder) and reported in ``missing``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch

from ._lib import CfgppError


# filled by solver_kwargs_from_dir: one line per CLIP tower that fell back from the HIP kernels to the torch-ops tower (callers and
# tests can assert on it; the GPU tests pin the HIP tower, so a silent fallback would be a different implementation)
last_text_tower_fallbacks: List[str] = []


def _first(*paths):
    for p in paths:
        if os.path.exists(p):
            return p
    return None


def _weights_in(folder, stem="diffusion_pytorch_model"):
    return _first(os.path.join(folder, stem + ".safetensors"), os.path.join(folder, stem + ".fp16.safetensors"))


def solver_kwargs_from_dir(model_dir, sdxl: bool, device="cuda", vae_dir=None, text_tower: str = "") -> Tuple[Dict, List[str]]:
    """(kwargs for ``get_solver``, names of the components not found).  ``vae_dir``: a folder holding the VAE weights,
    overriding the default lookup (SD1.5: ``<dir>/vae``; SDXL: ``<dir>/vae_fp16_fix`` or ``<dir>/sdxl-vae-fp16-fix`` -
    NEVER ``<dir>/vae``, whose fp16 activations overflow).  ``text_tower`` (or ``$CFGPP_TEXT_TOWER``): ``"hip"`` = the HIP CLIP
    tower or an error, ``"torch"`` = the torch-ops ``ClipTextTower`` even on a GPU; default = the HIP tower on a GPU, and when
    that one cannot take the checkpoint (an activation it has no kernel for, keys of a fine-tune it does not know) the torch-ops
    tower with a logged warning - the text encoder runs once per prompt and is not on the hot path."""
    import logging
    from .conditioning import ClipTextTower
    text_tower = (text_tower or os.environ.get("CFGPP_TEXT_TOWER", "")).lower()
    if text_tower not in ("", "hip", "torch"):
        raise ValueError(f"text_tower={text_tower!r}: expected 'hip', 'torch' or ''")
    d = str(model_dir)
    kw, missing = {}, []
    unet = _weights_in(os.path.join(d, "unet"))
    if vae_dir is not None:
        vae = _weights_in(str(vae_dir))
    elif sdxl:
        vae = _weights_in(os.path.join(d, "vae_fp16_fix")) or _weights_in(os.path.join(d, "sdxl-vae-fp16-fix"))
    else:
        vae = _weights_in(os.path.join(d, "vae"))
    if unet:
        kw["unet_weights"] = unet
    else:
        missing.append("unet")
    if vae:
        kw["vae_weights"] = vae
    else:
        missing.append("vae (SDXL needs madebyollin/sdxl-vae-fp16-fix in <dir>/vae_fp16_fix)" if sdxl else "vae")
    on_gpu = torch.device(device).type == "cuda"
    dtype = torch.float16 if on_gpu else torch.float32          # the reference runs the text encoders in the pipeline's fp16

    fallbacks = last_text_tower_fallbacks
    del fallbacks[:]

    def tower(enc, tok, **args):
        e, t = os.path.join(d, enc), os.path.join(d, tok)
        ok = os.path.exists(os.path.join(e, "config.json")) and _weights_in(e, "model") is not None and \
            all(os.path.exists(os.path.join(t, f)) for f in ("vocab.json", "merges.txt"))
        if not ok:
            missing.append(enc)
            return None
        if on_gpu and text_tower != "torch":
            from .text import HipClipTextTower
            try:
                return HipClipTextTower.from_dir(e, t, device=device, **args)
            except (CfgppError, KeyError, ValueError, NotImplementedError) as exc:
                # "this checkpoint is not one the HIP tower takes" (activation, missing / unexpected keys, shapes).  Anything else -
                # a missing library symbol, a HIP out-of-memory, a bug in the tower - propagates: it must not hide behind a fallback.
                if text_tower == "hip":
                    raise
                fallbacks.append(f"{enc}: torch-ops tower ({type(exc).__name__}: {exc})")
                logging.getLogger("cfgpp_amd").warning("HIP text tower cannot load %s (%s: %s); using the torch-ops ClipTextTower",
                                                       e, type(exc).__name__, exc)
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
