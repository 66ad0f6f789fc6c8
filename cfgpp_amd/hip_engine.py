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
from ._lib import CfgppError
from .unet_config import CONFIGS, UNetConfig
from .weights import load_safetensors_iter, synth_state_dict_iter


def _lane_spans(B: int, lanes: int):
    """Row spans [(r0, r1), ...] of the UNet batch [uc_1..uc_B, c_1..c_B] for `lanes` concurrent forwards.  A lane never
    straddles the uc / c boundary (its latents are then ONE contiguous slice of z): lanes = 2 is "null-prompt half,
    prompt half", lanes = 4 splits each half in two, and so on; lanes beyond 2 B collapse."""
    if lanes <= 1:
        return [(0, 2 * B)]
    per_half = max(1, min(lanes // 2, B))
    spans = []
    for half in range(2):
        for k in range(per_half):
            a, b = (k * B) // per_half, ((k + 1) * B) // per_half
            spans.append((half * B + a, half * B + b))
    return spans


class HipEngine:
    """``lanes`` (default: env CFGPP_LANES, else 1): the UNet batch of 2 B rows is split into that many row groups, each with
    its own UNet executor (own activation buffers; weights are loaded into each) on its own HIP stream.  The groups of one
    ``predict`` run CONCURRENTLY: consecutive launches of one stream never overlap on this hardware, so every launch's
    prologue, epilogue burst and tail leave CUs idle; with two or more streams the dispatcher fills those holes with the other
    lane's workgroups.  Ordering is by HIP events only (the caller's stream -> lanes -> the caller's stream): no host sync.
    Per-row results do not depend on which lane computed them beyond the rule-based K-split launches (whose tile count
    follows the row count)."""

    def __init__(self, cfg: UNetConfig, max_batch: int = 1, latent_hw: Optional[Tuple[int, int]] = None,
                 device=None, weights="synthetic", weight_seed: int = 0, lanes: Optional[int] = None):
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
        if lanes is None:
            lanes = int(os.environ.get("CFGPP_LANES", "1"))
        if lanes < 1 or (lanes > 1 and lanes % 2):
            raise CfgppError(f"lanes={lanes}: 1 or an even number (a lane never straddles the null-prompt / prompt halves)")
        # One device per process (the torchrun layout; cfgpp_unet_create / cfgpp_vae_create refuse a second one): the engine's
        # device BECOMES the process's current device and stays it - cfgpp_unet_finalize, the lazily allocated K-split
        # workspaces and every launch run against the current device, and none of the C entry points carries a device guard.
        torch.cuda.set_device(self.device)
        self.lanes = len(_lane_spans(self.max_batch, lanes))
        self._lanes_req = lanes
        lane_rows = max(b - a for a, b in _lane_spans(self.max_batch, lanes))
        self.units = [E.HipUNet(cfg, max_rows=lane_rows, sample_hw=latent_hw, device=self.device.index) for _ in range(self.lanes)]
        self.unet = self.units[0]
        if weights == "synthetic":
            items = synth_state_dict_iter(cfg, weight_seed)
        elif isinstance(weights, str):
            items = load_safetensors_iter(weights)
        else:
            items = weights.items() if isinstance(weights, dict) else weights
        for k, v in items:
            for u in self.units:
                u.load_tensor(k, v)
        for u in self.units:
            u.finalize()
        self.H, self.W = self.unet.H, self.unet.W
        self._streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)] if self.lanes > 1 else []
        self._ev_in = torch.cuda.Event() if self.lanes > 1 else None
        self._ev_out = [torch.cuda.Event() for _ in range(self.lanes)] if self.lanes > 1 else []
        self._spans = [(0, 0)]
        self._pins_shared = {}        # lane row count -> True once lane 0's tile pins were handed to the other lanes
        self._ctx_key = None
        self._eps = None
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
        self.B = B
        self._eps = torch.empty((2 * B, self.cfg.out_channels, self.H, self.W), dtype=torch.float16, device=self.device)
        if self.lanes == 1:
            self._spans = [(0, 2 * B)]
            self.unet.set_context(ehs, text_embeds, time_ids)
            return
        self._spans = _lane_spans(B, self._lanes_req)
        cur = torch.cuda.current_stream(self.device)
        ehs = ehs.to(device=self.device, dtype=torch.float16).contiguous()
        self._ev_in.record(cur)
        for (r0, r1), u, st, ev in zip(self._spans, self.units, self._streams, self._ev_out):
            te, ti = text_embeds, time_ids
            if te is not None and int(te.shape[0]) != 1:
                te, ti = te[r0:r1], ti[r0:r1]
            st.wait_event(self._ev_in)
            with torch.cuda.stream(st):
                u.set_context(ehs[r0:r1], te, ti)
            ev.record(st)
            cur.wait_event(ev)

    def _share_pins(self, rows: int):
        """lane 0 has just run its first forward at `rows` rows (the in-situ tile tuner ran inside it, alone on the GPU): the other
        lanes of the same size take its pins instead of tuning while lane 0 keeps the GPU busy"""
        if self._pins_shared.get(rows):
            return
        self._pins_shared[rows] = True
        try:
            hints = self.unet.export_tuning(rows)
        except CfgppError:
            return                      # tuner off (forced tile config): nothing to share
        for (r0, r1), u in zip(self._spans[1:], self.units[1:]):
            if r1 - r0 == rows:
                u.import_tuning(hints, rows)

    def predict(self, z: torch.Tensor, t: float):
        """(eps_uc, eps_c), each [B,4,H,W] fp16 - replaces predict_noise's UNet call + chunk(2)."""
        B = self.B
        if self.lanes == 1:
            eps = self.unet.forward(z, float(t), self._eps)
            if self.check_finite:
                self._finite_or_raise(t)
            return eps[:B], eps[B:]
        cur = torch.cuda.current_stream(self.device)
        self._ev_in.record(cur)
        for i, ((r0, r1), u, st, ev) in enumerate(zip(self._spans, self.units, self._streams, self._ev_out)):
            half = 0 if r1 <= B else 1
            a, b = r0 - half * B, r1 - half * B                # latent rows of this lane
            st.wait_event(self._ev_in)
            with torch.cuda.stream(st):
                u.forward(z[a:b], float(t), self._eps[r0:r1])
            if i == 0:
                self._share_pins(r1 - r0)
            ev.record(st)
            cur.wait_event(ev)
        if self.check_finite:
            self._finite_or_raise(t)
        return self._eps[:B], self._eps[B:]

    # -- tile pins of the whole engine (bench.py: rank 0 tunes, every rank imports) ----
    def export_tuning(self):
        return self.unet.export_tuning(self._spans[0][1] - self._spans[0][0])

    def import_tuning(self, hints, batch: int):
        spans = _lane_spans(int(batch), self._lanes_req)
        for (r0, r1), u in zip(spans, self.units):
            if r1 - r0 == spans[0][1] - spans[0][0]:
                u.import_tuning(hints, r1 - r0)
        self._pins_shared[spans[0][1] - spans[0][0]] = True

    def device_bytes(self) -> float:
        return sum(u.device_bytes() for u in self.units)

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
        return self.unet.flops(rows)          # (linear in rows: the lane split does not change the algorithmic work)
