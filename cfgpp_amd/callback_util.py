"""Per-step callback protocol of the reference (utils/callback_util.py:6-74).

``callback_fn(step, t, {'z0t', 'zt', 'decode'}) -> dict`` and the returned
``z0t`` / ``zt`` REPLACE the loop state (latent_diffusion.py:668-674).  The base
class fires when ``(step+1) % frequency == 0 or step == 0`` (utils/callback_util.py:32).
PNG writing is plumbing: torchvision is absent here, so images are written as
PNG through PIL when available and as ``.npy`` otherwise.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

__CALLBACK__ = {}


def register_callback(name):
    def wrapper(cls):
        if __CALLBACK__.get(name) is not None:
            raise NameError(f"Callback {name} is already registered")
        __CALLBACK__[name] = cls
        return cls
    return wrapper


def get_callback(name, **kwargs):
    if __CALLBACK__.get(name) is None:
        raise NameError(f"Callback {name} is not registered")
    return __CALLBACK__[name](**kwargs)


def save_image(img: torch.Tensor, path: Path):
    """img [B,3,H,W] in [0,1] -> one file per batch element (suffix _b when B > 1)."""
    arr = (img.detach().float().clamp(0, 1).cpu().numpy() * 255.0 + 0.5).astype(np.uint8)
    path = Path(path)
    for b in range(arr.shape[0]):
        p = path if arr.shape[0] == 1 else path.with_name(f"{path.stem}_{b}{path.suffix}")
        try:
            from PIL import Image
            Image.fromarray(arr[b].transpose(1, 2, 0)).save(p)
        except Exception:  # noqa: BLE001 - PIL missing: keep the data
            np.save(str(p) + ".npy", arr[b])


class DiffusionCallback:
    def __init__(self, frequency: int, workdir: Path):
        assert frequency > 0, "Frequency must be a positive float"
        self.frequency = frequency
        self.workdir = Path(workdir)

    def __call__(self, step, t, callback_kwargs):
        if (step + 1) % self.frequency == 0 or step == 0:
            return self.callback(step, t, callback_kwargs)
        return callback_kwargs

    def callback(self, step, t, callback_kwargs):
        raise NotImplementedError


@register_callback("draw_tweedie")
class DrawTweedieCallback(DiffusionCallback):
    def __init__(self, frequency: int, workdir: Path):
        super().__init__(frequency, workdir)
        self.workdir.joinpath("record/tweedie").mkdir(parents=True, exist_ok=True)

    @torch.no_grad()
    def callback(self, step, t, callback_kwargs):
        x0t = callback_kwargs["decode"](callback_kwargs["z0t"])
        x0t = (x0t / 2 + 0.5).clamp(0, 1).cpu()
        save_image(x0t, self.workdir.joinpath(f"record/tweedie/x0_{int(t)}.png"))
        return callback_kwargs


@register_callback("draw_noisy")
class DrawNoisyCallback(DiffusionCallback):
    def __init__(self, frequency: int, workdir: Path):
        super().__init__(frequency, workdir)
        self.workdir.joinpath("record/noisy").mkdir(parents=True, exist_ok=True)

    @torch.no_grad()
    def callback(self, step, t, callback_kwargs):
        xt = callback_kwargs["decode"](callback_kwargs["zt"])
        xt = (xt / 2 + 0.5).clamp(0, 1).cpu()
        save_image(xt, self.workdir.joinpath(f"record/noisy/xt_{int(t)}.png"))
        return callback_kwargs


class ComposeCallback(DiffusionCallback):
    def __init__(self, workdir, callbacks, frequency: int = 5):
        super().__init__(frequency, workdir)
        self.callbacks = [get_callback(name, workdir=Path(workdir), frequency=frequency) for name in callbacks]

    def __call__(self, step, t, callback_kwargs):
        for callback in self.callbacks:
            callback_kwargs = callback(step, t, callback_kwargs)
        return callback_kwargs
