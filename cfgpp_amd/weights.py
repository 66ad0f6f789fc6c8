"""Seeded synthetic UNet weights in the exact diffusers shapes (there are no real
checkpoints and no network in the build / GPU boxes), plus a safetensors loader
for real ``unet/diffusion_pytorch_model*.safetensors`` files.

Initialisation is variance preserving (fan-in scaled) with damped residual
branches so that a 50-step chain through the random network stays well inside
fp16 range; values are rounded through fp16 so the fp32 CPU oracle and the fp16
HIP engine consume bit-identical parameters.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Iterator, Tuple

import torch

from .unet_config import UNetConfig, param_shapes


def _seed_for(key: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_for(key, seed))
    leaf = key.rsplit(".", 1)[-1]
    is_norm = ".norm" in key or key.startswith("conv_norm_out")
    if is_norm:
        if leaf == "weight":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
    elif leaf == "bias":
        t = 0.02 * torch.randn(shape, generator=g)
    else:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        gain = 1.0
        # damp the branches that write into the residual stream
        if any(s in key for s in (".conv2.", ".to_out.0.", ".ff.net.2.", ".proj_out.")):
            gain = 0.5
        if key.startswith("conv_out"):
            gain = 1.0
        t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
    return t.to(torch.float16).to(torch.float32)      # fp16-representable fp32


def synth_state_dict_iter(cfg: UNetConfig, seed: int = 0) -> Iterator[Tuple[str, torch.Tensor]]:
    for key, shape in param_shapes(cfg).items():
        yield key, synth_tensor(key, shape, seed)


def synth_state_dict(cfg: UNetConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return dict(synth_state_dict_iter(cfg, seed))


def load_safetensors_iter(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    """Stream a diffusers UNet safetensors file key by key."""
    from safetensors import safe_open
    with safe_open(path, framework="pt", device="cpu") as f:
        for k in f.keys():
            yield k, f.get_tensor(k)
