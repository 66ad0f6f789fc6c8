"""Python face of the HIP engine: PyTorch-ROCm tensors in, raw HIP kernels
underneath (ctypes -> libcfgpp_hip.so).  Torch is plumbing only (device
memory, streams); no torch op computes anything on the hot path.

``HipUNet``      replaces ``pipe.unet`` of the reference
                 (latent_diffusion.py:63-67,146-156; latent_sdxl.py:40,50,170-183).
``step_ddim`` /  replace the per-step sampler arithmetic
``step_kdiff``   (latent_diffusion.py:660-666 etc.).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import CfgppError, UNetConfigC, check
from .unet_config import UNetConfig


def _require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise CfgppError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise CfgppError(f"{name} must be contiguous")


def _stream_ptr(t: Optional[torch.Tensor] = None) -> int:
    """the current torch stream of the device the tensor lives on (not of whatever device is current)"""
    if t is not None and t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------
# sampler step kernels
# ----------------------------------------------------------------------------
def step_ddim(z: torch.Tensor, z0t_out: torch.Tensor, eps_uc: torch.Tensor, eps_c: torch.Tensor,
              lam: float, coeffs: Tuple[float, float, float, float], tweedie_uc: bool, renoise_uc: bool):
    """In-place generalised DDIM update (see include/cfgpp.h: cfgpp_step_ddim)."""
    lib = _lib.load()
    for n, t in (("z", z), ("z0t_out", z0t_out), ("eps_uc", eps_uc), ("eps_c", eps_c)):
        _require_cuda(t, n)
    if z.dtype != z0t_out.dtype or z.dtype not in (torch.float16, torch.float32):
        raise CfgppError("step_ddim: z and z0t_out must both be fp32 or both fp16")
    if eps_uc.dtype != eps_c.dtype or eps_uc.dtype not in (torch.float16, torch.float32):
        raise CfgppError("step_ddim: eps must both be fp16 or both fp32")
    if not (z.numel() == z0t_out.numel() == eps_uc.numel() == eps_c.numel()):
        raise CfgppError("step_ddim: size mismatch")
    c1, c2, c3, c4 = (float(x) for x in coeffs)
    if z.dtype == torch.float16:          # fp16 latent (inversion / edit paths): every op rounds to fp16
        if eps_uc.dtype != torch.float16:
            raise CfgppError("step_ddim: an fp16 latent needs fp16 eps")
        check(lib.cfgpp_step_ddim_h(z.data_ptr(), z0t_out.data_ptr(), eps_uc.data_ptr(), eps_c.data_ptr(), float(lam),
                                    c1, c2, c3, c4, int(bool(tweedie_uc)), int(bool(renoise_uc)), z.numel(), _stream_ptr(z)),
              "cfgpp_step_ddim_h")
        return
    check(lib.cfgpp_step_ddim(z.data_ptr(), z0t_out.data_ptr(), eps_uc.data_ptr(), eps_c.data_ptr(),
                              1 if eps_uc.dtype == torch.float16 else 0, float(lam), c1, c2, c3, c4,
                              int(bool(tweedie_uc)), int(bool(renoise_uc)), z.numel(), _stream_ptr(z)),
          "cfgpp_step_ddim")


def kdiff_input(x: torch.Tensor, xc_out: torch.Tensor, s: float, mode: int):
    lib = _lib.load()
    _require_cuda(x, "x")
    _require_cuda(xc_out, "xc_out")
    if x.dtype != torch.float16 or xc_out.dtype != torch.float16:
        raise CfgppError("kdiff_input: fp16 latents expected")
    check(lib.cfgpp_kdiff_input(x.data_ptr(), xc_out.data_ptr(), float(s), int(mode), x.numel(), _stream_ptr(x)),
          "cfgpp_kdiff_input")


def step_kdiff(x: torch.Tensor, den_out: torch.Tensor, old: Optional[torch.Tensor], eps_uc: torch.Tensor,
               eps_c: torch.Tensor, coef9, variant: int, xl_form: bool, euler_branch: bool, write_old: bool):
    lib = _lib.load()
    for n, t in (("x", x), ("den_out", den_out), ("eps_uc", eps_uc), ("eps_c", eps_c)):
        _require_cuda(t, n)
        if t.dtype != torch.float16:
            raise CfgppError(f"step_kdiff: {n} must be fp16")
    if old is not None:
        _require_cuda(old, "old")
    arr = (C.c_float * 9)(*[float(v) for v in coef9])
    check(lib.cfgpp_step_kdiff(x.data_ptr(), den_out.data_ptr(), _ptr(old), eps_uc.data_ptr(), eps_c.data_ptr(),
                               arr, int(variant), int(bool(xl_form)), int(bool(euler_branch)), int(bool(write_old)),
                               x.numel(), _stream_ptr(x)), "cfgpp_step_kdiff")


def kdiff_denoise(x, eps_uc, eps_c, lam: float, sigma: float, den_out, uden_out):
    lib = _lib.load()
    for n, t in (("x", x), ("eps_uc", eps_uc), ("eps_c", eps_c), ("den_out", den_out), ("uden_out", uden_out)):
        _require_cuda(t, n)
        if t.dtype != torch.float16:
            raise CfgppError(f"kdiff_denoise: {n} must be fp16")
    check(lib.cfgpp_kdiff_denoise(x.data_ptr(), eps_uc.data_ptr(), eps_c.data_ptr(), float(lam), float(sigma),
                                  den_out.data_ptr(), uden_out.data_ptr(), x.numel(), _stream_ptr(x)), "cfgpp_kdiff_denoise")


def lincomb(out, x, y, z, a: float, b: float, mode: int):
    lib = _lib.load()
    for n, t in (("out", out), ("x", x), ("y", y)) + ((("z", z),) if z is not None else ()):
        _require_cuda(t, n)
        if t.dtype != torch.float16:
            raise CfgppError(f"lincomb: {n} must be fp16")
    check(lib.cfgpp_lincomb(out.data_ptr(), x.data_ptr(), y.data_ptr(), _ptr(z), float(a), float(b), int(mode), x.numel(),
                            _stream_ptr(x)), "cfgpp_lincomb")


# ----------------------------------------------------------------------------
# UNet engine
# ----------------------------------------------------------------------------
class HipUNet:
    """Hand-written HIP UNet behind the C ABI.  One instance per device."""

    def __init__(self, cfg: UNetConfig, max_rows: int, sample_hw: Optional[Tuple[int, int]] = None,
                 device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise CfgppError("HipUNet needs a ROCm GPU (torch.cuda.is_available() is False); "
                             "there is no CPU fallback for the HIP path")
        self.cfg = cfg
        self.device = int(device)
        self.max_rows = int(max_rows)
        H, W = sample_hw if sample_hw is not None else (cfg.sample_size, cfg.sample_size)
        self.H, self.W = int(H), int(W)
        cc = UNetConfigC()
        cc.in_channels, cc.out_channels = cfg.in_channels, cfg.out_channels
        cc.num_levels = cfg.num_levels
        for i in range(cfg.num_levels):
            cc.block_out_channels[i] = cfg.block_out_channels[i]
            cc.level_has_attn[i] = cfg.level_has_attn[i]
            cc.transformer_depth[i] = cfg.transformer_depth[i]
            cc.num_heads[i] = cfg.num_heads[i]
        cc.layers_per_block = cfg.layers_per_block
        cc.cross_attention_dim = cfg.cross_attention_dim
        cc.addition_embed = cfg.addition_embed
        cc.addition_time_embed_dim = cfg.addition_time_embed_dim
        cc.addition_pooled_dim = cfg.addition_pooled_dim
        cc.norm_groups = cfg.norm_groups
        cc.sample_h, cc.sample_w = self.H, self.W
        cc.max_rows = self.max_rows
        self._h = self.lib.cfgpp_unet_create(C.byref(cc), self.device)
        if not self._h:
            raise CfgppError("cfgpp_unet_create failed: " + _lib.last_error())
        self._keep = {}          # tensors the engine holds raw pointers to
        self.finalized = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.lib.cfgpp_unet_destroy(h)
            self._h = None

    # -- weights ---------------------------------------------------------------
    def load_tensor(self, key: str, t: torch.Tensor):
        t = t.detach().cpu().contiguous()
        if t.dtype == torch.float16:
            dt = 1
        else:
            t = t.to(torch.float32)
            dt = 0
        shape = (C.c_long * t.dim())(*t.shape)
        check(self.lib.cfgpp_unet_load_tensor(self._h, key.encode(), t.data_ptr(), dt, shape, t.dim()),
              f"cfgpp_unet_load_tensor({key})")

    def load_state_dict(self, items: Iterable[Tuple[str, torch.Tensor]]):
        if isinstance(items, dict):
            items = items.items()
        for k, v in items:
            self.load_tensor(k, v)
        return self

    def finalize(self):
        check(self.lib.cfgpp_unet_finalize(self._h), "cfgpp_unet_finalize")
        self.finalized = True
        return self

    # -- conditioning ------------------------------------------------------------
    def set_context(self, ehs: torch.Tensor, text_embeds: Optional[torch.Tensor] = None,
                    time_ids: Optional[torch.Tensor] = None):
        """ehs [rows,77,D] (uc rows first, then c rows).  SDXL: text_embeds [rows|1,1280], time_ids [rows|1,6]."""
        dev = torch.device("cuda", self.device)
        ehs = ehs.to(device=dev, dtype=torch.float16).contiguous()
        rows, tokens = int(ehs.shape[0]), int(ehs.shape[1])
        te = ti = None
        cond_rows = 0
        if self.cfg.addition_embed:
            if text_embeds is None or time_ids is None:
                raise CfgppError("set_context: SDXL needs text_embeds and time_ids")
            te = text_embeds.to(device=dev, dtype=torch.float16).contiguous()
            ti = time_ids.to(device=dev, dtype=torch.float32).contiguous()
            cond_rows = int(te.shape[0])
            if int(ti.shape[0]) != cond_rows:
                raise CfgppError("set_context: text_embeds / time_ids row mismatch")
        self._keep["ctx"] = (ehs, te, ti)
        self.rows = rows
        check(self.lib.cfgpp_unet_set_context(self._h, ehs.data_ptr(), rows, tokens, _ptr(te), _ptr(ti), cond_rows,
                                              _stream_ptr(ehs)), "cfgpp_unet_set_context")

    # -- forward -----------------------------------------------------------------
    def forward(self, z: torch.Tensor, t: float, eps_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eps[rows,4,H,W] fp16 = UNet(z[row % z_rows], t); rows were fixed by set_context."""
        _require_cuda(z, "z")
        if z.dtype not in (torch.float16, torch.float32):
            raise CfgppError("forward: z must be fp16 or fp32")
        zr = int(z.shape[0])
        if tuple(z.shape[1:]) != (self.cfg.in_channels, self.H, self.W):
            raise CfgppError(f"forward: z shape {tuple(z.shape)} != [*, {self.cfg.in_channels}, {self.H}, {self.W}]")
        rows = self.rows
        if rows % zr != 0:
            raise CfgppError(f"forward: rows={rows} is not a multiple of z rows={zr}")
        if eps_out is None:
            eps_out = torch.empty((rows, self.cfg.out_channels, self.H, self.W), dtype=torch.float16, device=z.device)
        check(self.lib.cfgpp_unet_forward(self._h, z.data_ptr(), 1 if z.dtype == torch.float16 else 0, zr, float(t),
                                          eps_out.data_ptr(), rows, _stream_ptr(z)), "cfgpp_unet_forward")
        return eps_out

    def sample_graph_ddim(self, z: torch.Tensor, z0t: torch.Tensor, eps: torch.Tensor, eps_uc: torch.Tensor, eps_c: torch.Tensor,
                          steps, lam: float, tweedie_uc: bool, renoise_uc: bool):
        """the whole DDIM loop as hipGraph replays (include/cfgpp.h: cfgpp_sample_graph_ddim).  ``steps``: per step
        ``(t, c1, c2, c3, c4)`` - what the eager loop passes to ``forward`` and ``step_ddim``; z is updated in place."""
        for n, t in (("z", z), ("z0t", z0t), ("eps", eps)):
            _require_cuda(t, n)
        if z.dtype != z0t.dtype or z.dtype not in (torch.float16, torch.float32) or eps.dtype != torch.float16:
            raise CfgppError("sample_graph_ddim: z / z0t must both be fp32 or both fp16, eps fp16")
        if tuple(z.shape[1:]) != (self.cfg.in_channels, self.H, self.W) or z.shape != z0t.shape:
            raise CfgppError(f"sample_graph_ddim: z shape {tuple(z.shape)}")
        if int(eps.shape[0]) != self.rows or self.rows % int(z.shape[0]) != 0:
            raise CfgppError(f"sample_graph_ddim: eps rows {int(eps.shape[0])} / z rows {int(z.shape[0])} vs context rows {self.rows}")
        flat = [float(v) for st in steps for v in st]
        if len(flat) != 5 * len(steps) or not steps:
            raise CfgppError("sample_graph_ddim: steps must be a non-empty list of (t, c1, c2, c3, c4)")
        arr = (C.c_float * len(flat))(*flat)
        check(self.lib.cfgpp_sample_graph_ddim(self._h, z.data_ptr(), z0t.data_ptr(), 1 if z.dtype == torch.float16 else 0, int(z.shape[0]),
                                               eps.data_ptr(), eps_uc.data_ptr(), eps_c.data_ptr(), self.rows, arr, len(steps), float(lam),
                                               int(bool(tweedie_uc)), int(bool(renoise_uc)), _stream_ptr(z)), "cfgpp_sample_graph_ddim")

    def profile(self, z: torch.Tensor, t: float, detail: bool = False) -> dict:
        """One forward with HIP events between launches: per kernel family ms / algorithmic flops / launches."""
        _require_cuda(z, "z")
        rows = self.rows
        eps = torch.empty((rows, self.cfg.out_channels, self.H, self.W), dtype=torch.float16, device=z.device)
        ms, fl, ln = (C.c_double * 4)(), (C.c_double * 4)(), (C.c_int * 4)()
        buf = C.create_string_buffer(1 << 20) if detail else None
        check(self.lib.cfgpp_unet_profile(self._h, z.data_ptr(), 1 if z.dtype == torch.float16 else 0, int(z.shape[0]),
                                          float(t), eps.data_ptr(), rows, _stream_ptr(z), ms, fl, ln, buf, (1 << 20) if detail else 0),
              "cfgpp_unet_profile")
        names = ("igemm", "attention", "norm", "small")
        out = {n: dict(ms=ms[i], flops=fl[i], launches=ln[i]) for i, n in enumerate(names)}
        if detail:
            out["detail"] = buf.value.decode()
        return out

    def export_tuning(self, rows: Optional[int] = None):
        """tile config pinned per igemm launch by the in-situ tuning at this batch (list of ints, plan order)"""
        rows = self.rows if rows is None else int(rows)
        buf = (C.c_int * 4096)()
        n = self.lib.cfgpp_unet_tuning(self._h, rows, buf, 4096, 0)
        if n < 0:
            raise CfgppError("cfgpp_unet_tuning: " + _lib.last_error())
        return [int(buf[i]) for i in range(n)]

    def import_tuning(self, hints, rows: int):
        buf = (C.c_int * len(hints))(*[int(h) for h in hints])
        n = self.lib.cfgpp_unet_tuning(self._h, int(rows), buf, len(hints), 1)
        if n < 0:
            raise CfgppError("cfgpp_unet_tuning: " + _lib.last_error())

    def flops(self, rows: int) -> float:
        return float(self.lib.cfgpp_unet_flops(self._h, int(rows)))

    def device_bytes(self) -> float:
        return float(self.lib.cfgpp_unet_device_bytes(self._h))
