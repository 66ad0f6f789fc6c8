"""Host-side scheduler tables for the CFG++ sampling loop.

Everything the reference obtains from ``diffusers`` schedulers plus its own
index conventions is restated here as plain fp32 tables that live on the HOST.
The HIP loop never synchronises with the host: every per-step coefficient is
looked up here before the step is enqueued.

Reference behaviour that is reproduced (file:line in /root/reference):

* ``latent_diffusion.py:69-80`` / ``latent_sdxl.py:56-67`` - DDIM table
  construction, ``skip = 1000 // NFE``, ``final_alpha_cumprod = abar[0]`` and
  the **shift by one**: ``alphas_cumprod = cat([1.0], abar)`` (quirk Q1).
* ``latent_diffusion.py:88-90`` - ``alpha(t)``: shifted table for ``t >= 0``,
  ``final_alpha_cumprod`` otherwise.
* ``latent_sdxl.py:732-734`` - SDXL DDIM solvers index the shifted table
  directly, so a negative ``t - skip`` wraps around (quirk Q3).
* ``latent_sdxl.py:407-418`` - Lightning: Euler "trailing" timesteps, no
  ``final_alpha_cumprod``.
* ``latent_diffusion.py:44-50`` - Karras sigma ramp.

The diffusers scheduler constants (scaled-linear betas 0.00085..0.012, 1000
train steps, ``steps_offset=1``, "leading" spacing for DDIM, "trailing" for
the Lightning Euler scheduler) are the published SD1.5 / SDXL scheduler
configs (diffusers 0.27.1, pinned by the reference's environment.yaml:87).
"""
from __future__ import annotations

import os

import numpy as np
import torch

NUM_TRAIN_TIMESTEPS = 1000
BETA_START = 0.00085
BETA_END = 0.012


def alphas_cumprod_computed() -> torch.Tensor:
    """abar[0..999], fp32, by diffusers' ``scaled_linear`` formula.  NOT bit-stable
    across hosts: ``torch.linspace`` differs by 1 ulp between CPU ISAs (observed
    between the build container and the MI355X box), which flips fp32 sampler
    outputs.  Use :func:`alphas_cumprod`."""
    betas = torch.linspace(BETA_START ** 0.5, BETA_END ** 0.5, NUM_TRAIN_TIMESTEPS,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


_ALPHA_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "alphas_cumprod_f32.npy")


def alphas_cumprod() -> torch.Tensor:
    """abar[0..999], fp32: the pinned table (``data/alphas_cumprod_f32.npy``) = the
    values the reference's scheduler produced when the golden vectors were
    recorded (tests/golden G1/total_alphas), identical on every host."""
    return torch.from_numpy(np.load(_ALPHA_FILE).astype(np.float32)).clone()


_SQRT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ddim_sqrt_tables.npz")


def ddim_sqrt_tables():
    """(sqrt(a), sqrt(1-a)) over the SHIFTED table ``cat([1.0], abar)`` (1001 entries), fp32.
    Pinned data, because ``torch.sqrt`` on fp32 is not bit-stable across hosts either
    (Sleef AVX2 vs AVX-512 kernels differ by 1 ulp on a few entries - observed between
    the build container and the MI355X box).  The file holds what the reference's
    ``at.sqrt()`` / ``(1-at).sqrt()`` evaluated to when the golden vectors were recorded."""
    d = np.load(_SQRT_FILE)
    return torch.from_numpy(d["sqrt_a"].astype(np.float32)).clone(), torch.from_numpy(d["sqrt_1ma"].astype(np.float32)).clone()


def ddim_timesteps(num_sampling: int, steps_offset: int = 1) -> torch.Tensor:
    """DDIMScheduler.set_timesteps, "leading" spacing -> int64 tensor."""
    step_ratio = NUM_TRAIN_TIMESTEPS // num_sampling
    ts = (np.arange(0, num_sampling) * step_ratio).round()[::-1].copy().astype(np.int64)
    return torch.from_numpy(ts + steps_offset)


def euler_trailing_timesteps(num_sampling: int) -> torch.Tensor:
    """EulerDiscreteScheduler(timestep_spacing="trailing") -> fp32 tensor."""
    ts = np.round(np.arange(NUM_TRAIN_TIMESTEPS, 0, -NUM_TRAIN_TIMESTEPS / num_sampling)) - 1
    return torch.from_numpy(ts.astype(np.float32))


def append_zero(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0) -> torch.Tensor:
    """Karras et al. (2022) ramp with a trailing zero (latent_diffusion.py:44-50)."""
    ramp = torch.linspace(0, 1, n + 1)[:-1]
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas)


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """latent_diffusion.py:30-37."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


class SchedulerTables:
    """The numbers a solver instance owns after ``__init__``.

    ``kind``: ``"ddim"`` (SD1.5 / SDXL base) or ``"lightning"`` (SDXL-Lightning).
    """

    def __init__(self, num_sampling: int, kind: str = "ddim"):
        if kind not in ("ddim", "lightning"):
            raise ValueError(f"unknown scheduler kind {kind}")
        self.kind = kind
        self.num_sampling = int(num_sampling)
        self.total_alphas = alphas_cumprod()                       # abar, len 1000
        self.sigmas = (1 - self.total_alphas).sqrt() / self.total_alphas.sqrt()
        self.log_sigmas = self.sigmas.log()
        self.skip = NUM_TRAIN_TIMESTEPS // self.num_sampling
        if kind == "ddim":
            self.timesteps = ddim_timesteps(self.num_sampling)     # int64
            self.final_alpha_cumprod = self.total_alphas[0].clone()
        else:
            self.timesteps = euler_trailing_timesteps(self.num_sampling)  # fp32
            self.final_alpha_cumprod = None                        # latent_sdxl.py:417
        # the shifted table (quirk Q1)
        self.alphas_cumprod = torch.cat([torch.tensor([1.0]), self.total_alphas])
        self._sqrt_a, self._sqrt_1ma = ddim_sqrt_tables()

    # -- reference index rules -------------------------------------------------
    def alpha(self, t) -> torch.Tensor:
        """``alpha(t)`` with the ``t < 0`` guard (latent_diffusion.py:88-90)."""
        t = int(t)
        if t >= 0:
            return self.alphas_cumprod[t]
        if self.final_alpha_cumprod is None:
            raise AttributeError("final_alpha_cumprod")   # what the reference would raise
        return self.final_alpha_cumprod

    def alpha_wrap(self, t) -> torch.Tensor:
        """Unguarded index used by the SDXL DDIM loops (latent_sdxl.py:732-734)."""
        return self.alphas_cumprod[int(t)]

    def _index(self, t, wrap: bool) -> int:
        """index into the shifted table for ``alpha(t)`` (guarded) or ``alphas_cumprod[t]`` (wrap)."""
        t = int(t)
        if wrap:
            return t % len(self.alphas_cumprod)          # python negative indexing
        if t >= 0:
            return t
        if self.final_alpha_cumprod is None:
            raise AttributeError("final_alpha_cumprod")
        return 1                                         # final_alpha_cumprod = abar[0] = shifted[1]

    def ddim_sqrt_coeffs(self, t, wrap: bool = False, inversion: bool = False):
        """(c1, c2, c3, c4) = sqrt(1-a_tw), sqrt(a_tw), sqrt(a_rn), sqrt(1-a_rn) as python floats
        (exact fp32 values) for the step at timestep ``t``:
        forward   a_tw = alpha(t),        a_rn = alpha(t - skip)   (latent_diffusion.py:655-666)
        inversion a_tw = alpha(t - skip), a_rn = alpha(t)          (latent_diffusion.py:901-908)"""
        i_t, i_p = self._index(t, wrap), self._index(int(t) - self.skip, wrap)
        i_tw, i_rn = (i_p, i_t) if inversion else (i_t, i_p)
        return (float(self._sqrt_1ma[i_tw]), float(self._sqrt_a[i_tw]), float(self._sqrt_a[i_rn]), float(self._sqrt_1ma[i_rn]))

    # -- k-diffusion helpers ---------------------------------------------------
    def timestep(self, sigma: torch.Tensor) -> torch.Tensor:
        """argmin |log sigma - log sigma_table| (latent_diffusion.py:211-214)."""
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def sigma_to_t(self, sigma: torch.Tensor, quantize: bool = True) -> torch.Tensor:
        """k_diffusion sigma->t (latent_sdxl.py:333-346)."""
        total_sigmas = (1 - self.total_alphas).sqrt() / self.total_alphas.sqrt()
        dists = sigma - total_sigmas[:, None]
        if quantize:
            return dists.abs().argmin(dim=0).view(sigma.shape)
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=total_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = total_sigmas[low_idx], total_sigmas[high_idx]
        w = ((low - sigma) / (low - high)).clamp(0, 1)
        t = (1 - w) * low_idx + w * high_idx
        return t.view(sigma.shape)

    def karras_sigmas(self) -> torch.Tensor:
        total_sigmas = (1 - self.total_alphas).sqrt() / self.total_alphas.sqrt()
        return get_sigmas_karras(self.num_sampling, total_sigmas.min(), total_sigmas.max(), rho=7.0)
