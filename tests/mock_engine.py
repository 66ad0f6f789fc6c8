"""CPU stand-in for ``cfgpp_amd.hip_engine.HipEngine`` used ONLY by tests, so the
solver control flow (index rules, coefficient tables, callbacks, conditioning
assembly, batching) can be checked against the reference's golden vectors
without a GPU.  It is injected explicitly (``get_solver(..., engine=mock)``);
the product never constructs it.

The arithmetic emulates the HIP step kernels' *interface* (fp32 coefficients in,
explicit fp16 roundings) with the oracle's rounding primitives.
"""
from __future__ import annotations

import torch

from oracle import sampler as O

H, F = torch.float16, torch.float32


def _h(x):
    return x.to(H)


def _f(x):
    return x.to(F)


def _c(v):
    return torch.tensor(float(v), dtype=F)


def _div(x, c):
    """``x / c`` as the kernels form it (step_kernels.hip: div_as_ref): c > 0 IEEE division, c < 0 multiply by -c (the
    host-side fp32 reciprocal: torch's GPU ``div`` with a CPU-scalar divisor)"""
    return x * (-c) if float(c) < 0 else x / c


def emulate_step_ddim(z, z0t_out, eps_uc, eps_c, lam, coeffs, tweedie_uc, renoise_uc):
    """what cfgpp_step_ddim computes (step_kernels.hip: ddim_step_kernel)."""
    c1, c2, c3, c4 = (_c(v) for v in coeffs)
    hat = O.cfg_mix(eps_uc, eps_c, lam)
    A = eps_uc if tweedie_uc else hat
    B = eps_uc if renoise_uc else hat
    if z.dtype == H:        # fp16 latent: every op rounds to fp16 (ddim_step_h_kernel)
        r = lambda t: _f(_h(t))  # noqa: E731
        pa, pb = r(_f(A) * c1), r(_f(B) * c4)
        z0 = r(_div(r(_f(z) - pa), c2))
        zn = r(r(c3 * z0) + pb)
        z0t_out.copy_(_h(z0))
        z.copy_(_h(zn))
        return
    if eps_uc.dtype == H:
        pa, pb = _f(_h(_f(A) * c1)), _f(_h(_f(B) * c4))
    else:
        pa, pb = A * c1, B * c4
    z0 = _div(_f(z) - pa, c2)
    zn = c3 * z0 + pb
    z0t_out.copy_(z0)
    z.copy_(zn)


def emulate_kdiff_input(x, xc, s, mode):
    v = _div(_f(x), _c(s)) if mode == 0 else _f(x) * _c(s)
    xc.copy_(_h(v))


def emulate_step_kdiff(x, den_out, old, eps_uc, eps_c, coef, variant, xl_form, euler_branch, write_old):
    """what cfgpp_step_kdiff computes (step_kernels.hip: kdiff_step_kernel)."""
    lam, sigma, c_out_h, sigma_item, sigma_next, neg_exp, expm1, two_r, exp_mh = (_c(v) for v in coef)
    xv = _f(x)
    uc = _f(eps_uc)
    hat = _f(O.cfg_mix(eps_uc, eps_c, float(lam)))
    r = lambda t: _f(_h(t))  # noqa: E731
    if xl_form:
        den = r(xv + r(hat * c_out_h))
        uden = r(xv + r(uc * c_out_h))
    else:
        den = r(xv - r(hat * sigma))
        uden = r(xv - r(uc * sigma))
    d_from = den if variant == 0 else uden
    if euler_branch:
        d = r(_div(r(xv - d_from), sigma_item))
        xn = r(den + r(d * sigma_next))
    else:
        ov = _f(old)
        diff_a = uden if variant == 2 else den
        term1 = r(d_from * neg_exp)
        t2 = r(_div(r(r(diff_a - ov) * expm1), two_r))
        extra1 = r(term1 - t2)
        extra2 = r(xv * exp_mh)
        xn = r(r(den + extra1) + extra2)
    den_out.copy_(_h(den))
    if write_old:
        old.copy_(_h(den if variant == 0 else uden))
    x.copy_(_h(xn))


def emulate_kdiff_denoise(x, eps_uc, eps_c, lam, sigma, den_out, uden_out):
    r = lambda t: _f(_h(t))  # noqa: E731
    xv, uc = _f(x), _f(eps_uc)
    hat = _f(O.cfg_mix(eps_uc, eps_c, float(lam)))
    den_out.copy_(_h(xv - r(hat * _c(sigma))))
    uden_out.copy_(_h(xv - r(uc * _c(sigma))))


def emulate_lincomb(out, x, y, z, a, b, mode):
    r = lambda t: _f(_h(t))  # noqa: E731
    xv, yv = _f(x), _f(y)
    if mode == 0:
        v = r(xv * _c(a)) - r(yv * _c(b))
    elif mode == 1:
        v = r(yv - r(_f(z) * _c(b))) + r(xv * _c(a))
    else:
        v = xv + r(yv * _c(a))
    out.copy_(_h(v))


class MockEngine:
    """unet_fn(z_rows [2B,4,H,W], t float, ehs [2B,77,D], text_embeds|None, time_ids|None) -> eps [2B,4,H,W]"""

    device = torch.device("cpu")

    def __init__(self, unet_fn, latent_hw=(8, 8)):
        self.unet_fn = unet_fn
        self.H, self.W = latent_hw
        self.calls = []
        self.contexts = []

    def set_context(self, uc, c, text_embeds=None, time_ids=None):
        B = max(uc.shape[0], c.shape[0])
        self.B = B
        self.ehs = torch.cat([uc.expand(B, -1, -1), c.expand(B, -1, -1)], 0)
        self.te, self.ti = text_embeds, time_ids
        self.contexts.append(dict(rows=2 * B, te=None if text_embeds is None else text_embeds.clone(),
                                  ti=None if time_ids is None else time_ids.clone()))

    def predict(self, z, t):
        zz = torch.cat([z, z], 0)
        eps = self.unet_fn(zz, float(t), self.ehs, self.te, self.ti)
        self.calls.append(dict(t=float(t), z=z.clone(), eps=eps.clone(), z_dtype=z.dtype))
        return eps[: self.B].contiguous(), eps[self.B:].contiguous()

    step_ddim = staticmethod(emulate_step_ddim)
    kdiff_input = staticmethod(emulate_kdiff_input)
    step_kdiff = staticmethod(emulate_step_kdiff)
    kdiff_denoise = staticmethod(emulate_kdiff_denoise)
    lincomb = staticmethod(emulate_lincomb)

    def randn_like(self, x):
        return torch.randn_like(x)


class StubVAE:
    """Shape-correct stand-in for the VAE in CPU solver tests (trajectories are compared on latents):
    decode = nearest x8 of the first 3 latent channels / scale, encode = 8x8 mean pool * scale (+ a zero channel)."""

    def __init__(self, scale: float):
        self.scale = float(scale)

    def decode(self, zt):
        z = zt.float() / self.scale
        return torch.nn.functional.interpolate(z[:, :3], scale_factor=8.0, mode="nearest")

    def encode(self, x, **_):
        m = torch.nn.functional.avg_pool2d(x.float(), 8)
        return torch.cat([m, torch.zeros_like(m[:, :1])], 1) * self.scale
