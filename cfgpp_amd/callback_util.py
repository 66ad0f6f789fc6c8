"""Per-step callbacks for the sampling loops.

Protocol (what the reference's solvers implement, latent_diffusion.py:668-674 and
utils/callback_util.py:22-74): the loop calls ``fn(step, t, state)`` with
``state = {'z0t': ..., 'zt': ..., 'decode': callable}`` and CONTINUES FROM THE RETURNED
dict - a callback may replace ``z0t`` / ``zt``.  Gated callbacks run on step 0 and on every
``frequency``-th step after it.  The two stock callbacks decode a latent and write one PNG per
firing, named by the integer timestep, under ``<workdir>/record/{tweedie,noisy}/``.

Design here: one gate (``due``), one snapshot implementation parameterised by (state key, folder,
file stem), a name -> factory table, and composition as a left fold.  The public names of the reference
(``get_callback``, ``register_callback``, ``DiffusionCallback``, ``ComposeCallback``, ``draw_tweedie``,
``draw_noisy``, ``__CALLBACK__``) are kept so user code ports unchanged.  torchvision is not a dependency:
images go through PIL, or ``.npy`` when PIL is missing.
"""
from __future__ import annotations

import functools
from pathlib import Path
from typing import Callable, Dict, Iterable

import numpy as np
import torch

State = dict
__CALLBACK__: Dict[str, Callable[..., "DiffusionCallback"]] = {}      # name -> factory(frequency=, workdir=)


def register_callback(name: str):
    """Decorator: publish a callback class or factory function under ``name`` (duplicates -> NameError)."""
    def publish(factory):
        if name in __CALLBACK__:
            raise NameError(f"Callback {name} is already registered")
        __CALLBACK__[name] = factory
        return factory
    return publish


def get_callback(name: str, **kwargs) -> "DiffusionCallback":
    factory = __CALLBACK__.get(name)
    if factory is None:
        raise NameError(f"Callback {name} is not registered")
    return factory(**kwargs)


def due(step: int, frequency: int) -> bool:
    """The firing rule: first step, then every ``frequency``-th one (steps counted from 1)."""
    return step == 0 or (step + 1) % frequency == 0


def save_image(img: torch.Tensor, path, normalize: bool = False) -> None:
    """``img`` [B,3,H,W] with values in [0,1] -> 8-bit RGB file(s); batch element b > 0 gets the suffix ``_b``
    (a single image keeps the given name).  ``normalize=True`` stretches the tensor's [min, max] to [0, 1] first -
    what ``torchvision.utils.save_image(..., normalize=True)`` does to the results the reference's example scripts save
    (examples/text_to_img.py:56, inversion.py:55, text_to_mscoco.py:62); the draw_* callbacks save without it."""
    path = Path(path)
    img = img.detach().float()
    if normalize:
        low, high = float(img.min()), float(img.max())
        img = (img.clamp(low, high) - low) / max(high - low, 1e-5)
    pixels = img.clamp(0, 1).mul(255.0).add(0.5).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    for b, frame in enumerate(pixels):
        target = path if len(pixels) == 1 else path.with_name(f"{path.stem}_{b}{path.suffix}")
        try:
            from PIL import Image
        except ImportError:
            np.save(f"{target}.npy", frame)
        else:
            Image.fromarray(frame).save(target)


class DiffusionCallback:
    """Base of gated callbacks: subclasses implement ``callback(step, t, state) -> state``."""

    def __init__(self, frequency: int, workdir):
        assert frequency > 0, "Frequency must be a positive float"
        self.frequency, self.workdir = frequency, Path(workdir)

    def __call__(self, step, t, state: State) -> State:
        return self.callback(step, t, state) if due(step, self.frequency) else state

    def callback(self, step, t, state: State) -> State:
        raise NotImplementedError


class LatentSnapshot(DiffusionCallback):
    """Decode ``state[key]`` with the solver's ``decode`` and save it as
    ``<workdir>/record/<folder>/<stem>_<int(t)>.png``; the state passes through untouched."""

    def __init__(self, frequency: int, workdir, key: str, folder: str, stem: str):
        super().__init__(frequency, workdir)
        self.key, self.stem = key, stem
        self.out_dir = self.workdir / "record" / folder
        self.out_dir.mkdir(parents=True, exist_ok=True)

    def callback(self, step, t, state: State) -> State:
        with torch.no_grad():
            image = state["decode"](state[self.key])
        save_image(image * 0.5 + 0.5, self.out_dir / f"{self.stem}_{int(t)}.png")       # [-1,1] -> [0,1]
        return state


@register_callback("draw_tweedie")
def DrawTweedieCallback(frequency: int, workdir) -> LatentSnapshot:
    """x0 estimate (Tweedie) of every firing step -> record/tweedie/x0_<t>.png"""
    return LatentSnapshot(frequency, workdir, key="z0t", folder="tweedie", stem="x0")


@register_callback("draw_noisy")
def DrawNoisyCallback(frequency: int, workdir) -> LatentSnapshot:
    """noisy latent of every firing step -> record/noisy/xt_<t>.png"""
    return LatentSnapshot(frequency, workdir, key="zt", folder="noisy", stem="xt")


class ComposeCallback(DiffusionCallback):
    """Several registered callbacks by name, applied in order; each one sees the state the previous returned."""

    def __init__(self, workdir, callbacks: Iterable[str], frequency: int = 5):
        super().__init__(frequency, workdir)
        self.callbacks = tuple(get_callback(n, workdir=self.workdir, frequency=frequency) for n in callbacks)

    def __call__(self, step, t, state: State) -> State:
        return functools.reduce(lambda st, cb: cb(step, t, st), self.callbacks, state)
