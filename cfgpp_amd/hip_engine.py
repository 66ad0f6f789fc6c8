"""The engine object a Solver drives: UNet forward + fused step kernels on ONE GPU.

``HipEngine`` is the only engine the product path constructs.  Tests may inject
another object with the same methods (``tests/mock_engine.py`` runs the solver
control flow on CPU against the oracle); nothing in this package falls back to
it.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import engine as E
from . import _lib
from ._lib import CfgppError
from .tune_cache import PinCache, build_id
from .unet_config import CONFIGS, UNetConfig
from .weights import load_safetensors_iter, synth_state_dict_iter


class HipEngine:
    """One UNet executor on one HIP stream (the caller's current stream).  Round 4 measured the alternative - the UNet batch
    split into row groups on concurrent streams, so that one group's launch prologues / epilogue bursts overlap the other's
    K loops - on the MI355X: 2 streams 20.75 ms against 19.35 ms per SD1.5 forward at 16 rows, 4 streams 26.4 ms (SDXL: 37.5 ->
    41.8 / 56.5 ms; profiles/r04/ab/forward_ab_fuse_ln_x_lanes_*): halving M per launch costs more than the overlap returns,
    so there is one stream."""

    def __init__(self, cfg: UNetConfig, max_batch: int = 1, latent_hw: Optional[Tuple[int, int]] = None,
                 device=None, weights="synthetic", weight_seed: int = 0):
        if isinstance(cfg, str):
            cfg = CONFIGS[cfg]
        if not torch.cuda.is_available():
            raise CfgppError("HipEngine needs a ROCm GPU; the HIP path has no CPU fallback")
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise CfgppError(f"HipEngine cannot run on device '{dev}': the HIP path has no CPU fallback")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.cfg = cfg
        self.max_batch = int(max_batch)
        # One device per process (the torchrun layout; cfgpp_unet_create / cfgpp_vae_create refuse a second one): the engine's
        # device BECOMES the process's current device and stays it - cfgpp_unet_finalize, the lazily allocated K-split
        # workspaces and every launch run against the current device, and none of the C entry points carries a device guard.
        torch.cuda.set_device(self.device)
        self.unet = E.HipUNet(cfg, max_rows=2 * self.max_batch, sample_hw=latent_hw, device=self.device.index)
        if weights == "synthetic":
            items = synth_state_dict_iter(cfg, weight_seed)
        elif isinstance(weights, str):
            items = load_safetensors_iter(weights)
        else:
            items = weights.items() if isinstance(weights, dict) else weights
        self.unet.load_state_dict(items).finalize()
        self.H, self.W = self.unet.H, self.unet.W
        # tile pins persist across processes (tune_cache.py): only the first process on a box runs the in-situ tuning passes
        self._pins = PinCache(getattr(cfg, "name", type(cfg).__name__), (self.H, self.W), torch.cuda.get_device_properties(self.device).name,
                              build_id(_lib.LIB_PATH), lambda rows: self.unet.export_tuning(rows), lambda h, rows: self.unet.import_tuning(h, rows),
                              knobs=lambda: (int(self.unet.lib.cfgpp_igemm_tuner_state()),))
        self._ctx_key = None
        self._eps = None
        self._g_buf = None
        # opt-in guard for the first runs with a real checkpoint (real SDXL activations approach the fp16 maximum in the deep
        # blocks; the synthetic weights of the tests do not): scan eps after every forward - one host sync per step - and stop
        # with the timestep instead of decoding a NaN image
        self.check_finite = os.environ.get("CFGPP_CHECK_FINITE", "0") not in ("", "0")

    def _finite_or_raise(self, t):
        if not bool(torch.isfinite(self._eps).all()):
            bad = int((~torch.isfinite(self._eps)).sum())
            raise CfgppError(f"UNet output at t={t} holds {bad} non-finite values (fp16 overflow inside the UNet?) - CFGPP_CHECK_FINITE")

    # -- conditioning ------------------------------------------------------------
    def set_context(self, uc: torch.Tensor, c: torch.Tensor, text_embeds=None, time_ids=None):
        """uc, c: [B or 1, 77, D].  Rows are laid out [uc_1..uc_B, c_1..c_B]
        (the batched form of torch.cat([uc, c]), latent_diffusion.py:152)."""
        B = max(int(uc.shape[0]), int(c.shape[0]))
        if B > self.max_batch:
            raise CfgppError(f"batch {B} exceeds engine max_batch {self.max_batch}")
        if uc.shape[0] != B:
            uc = uc.expand(B, -1, -1)
        if c.shape[0] != B:
            c = c.expand(B, -1, -1)
        ehs = torch.cat([uc, c], dim=0)
        self.unet.set_context(ehs, text_embeds, time_ids)
        self._pins.load(2 * B)
        self.B = B
        if self._eps is None or int(self._eps.shape[0]) != 2 * B:      # kept across calls: a captured graph holds its address
            self._eps = torch.empty((2 * B, self.cfg.out_channels, self.H, self.W), dtype=torch.float16, device=self.device)

    def predict(self, z: torch.Tensor, t: float):
        """(eps_uc, eps_c), each [B,4,H,W] fp16 - replaces predict_noise's UNet call + chunk(2)."""
        eps = self.unet.forward(z, float(t), self._eps)
        self._pins.save(self.unet.rows)          # (no-op after the first forward at this batch)
        if self.check_finite:
            self._finite_or_raise(t)
        return eps[: self.B], eps[self.B:]

    # -- whole-loop graph replay ---------------------------------------------------------
    @property
    def graph_enabled(self) -> bool:
        """$CFGPP_GRAPH=1: DDIM loops without a callback run as hipGraph replays of one captured step (default off: see DESIGN.md)"""
        return os.environ.get("CFGPP_GRAPH", "0") not in ("", "0")

    def ddim_loop_graph(self, zt: torch.Tensor, steps, lam: float, tweedie_uc: bool, renoise_uc: bool, single: str = ""):
        """run the whole loop on (a persistent copy of) ``zt``; returns (z0t, zt) as fresh tensors.  ``single``: "uc" / "c" when the
        caller passed only one conditioning (both eps halves are then the same rows, like predict_noise's)"""
        key = (int(zt.shape[0]), zt.dtype)
        if self._g_buf is None or self._g_buf[0] != key:
            self._g_buf = (key, torch.empty_like(zt), torch.empty_like(zt))      # stable addresses: one capture serves every sample() call
        _, gz, gz0 = self._g_buf
        gz.copy_(zt)
        eps = self._eps
        euc, ec = eps[: self.B], eps[self.B:]
        if single == "uc":
            ec = euc
        elif single == "c":
            euc = ec
        self.unet.sample_graph_ddim(gz, gz0, eps, euc, ec, steps, lam, tweedie_uc, renoise_uc)
        self._pins.save(self.unet.rows)
        if self.check_finite:
            self._finite_or_raise(steps[-1][0])
        return gz0.clone(), gz.clone()

    # -- tile pins (bench.py: rank 0 tunes, every rank imports) ----
    def export_tuning(self):
        return self.unet.export_tuning(self.unet.rows)

    def import_tuning(self, hints, batch: int):
        self.unet.import_tuning(hints, 2 * int(batch))
        self._pins.mark_imported(2 * int(batch))       # an explicit import wins over whatever this rank's disk cache holds

    def device_bytes(self) -> float:
        return self.unet.device_bytes()

    # -- fused sampler arithmetic ---------------------------------------------------
    step_ddim = staticmethod(E.step_ddim)
    kdiff_input = staticmethod(E.kdiff_input)
    step_kdiff = staticmethod(E.step_kdiff)
    kdiff_denoise = staticmethod(E.kdiff_denoise)
    lincomb = staticmethod(E.lincomb)

    def randn_like(self, x):
        """ancestral noise: device RNG, like the reference's torch.randn_like on the GPU"""
        return torch.randn_like(x)

    def flops_per_forward(self, rows: int) -> float:
        return self.unet.flops(rows)
