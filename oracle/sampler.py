"""CPU oracle for the CFG++ sampler arithmetic.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``cfgpp_amd``) never does.

It restates, with every rounding written out, what the reference's sampling
loops compute once the UNet has returned ``(eps_uc, eps_c)``.  The reference
relies on PyTorch type promotion (fp16 eps from the autocast UNet, fp32 latent,
0-dim fp32 CPU scalars); here every intermediate has an explicit dtype so the
HIP kernel in ``cfgpp_amd/csrc/step_kernels.hip`` can be checked bit-for-bit.

Pinned against the golden vectors recorded from the reference itself
(``tests/golden/sampler_golden.npz``, made by ``tests/golden/make_golden.py``):
see ``tests/test_oracle_golden.py``.

Reference lines (in /root/reference):
  cfg mix                 latent_diffusion.py:660          latent_sdxl.py:738
  DDIM CFG / CFG++        latent_diffusion.py:283-286, 663-666   latent_sdxl.py:453-456, 741-744
  inversion CFG / CFG++   latent_diffusion.py:179-180, 907-908   latent_sdxl.py:317-318, 972-973
  k-diffusion helpers     latent_diffusion.py:216-241      latent_sdxl.py:326-363
  DPM++2M CFG / CFG++     latent_diffusion.py:479-490, 855-866   latent_sdxl.py:904-919
  Euler  CFG / CFG++      latent_diffusion.py:331-333, 708-710   latent_sdxl.py:503-506, 794-797
"""
from __future__ import annotations

import torch

H = torch.float16
F = torch.float32


def _h(x: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to fp16 (what a fp16 result tensor stores)."""
    return x.to(H)


def _f(x: torch.Tensor) -> torch.Tensor:
    return x.to(F)


def _s(v) -> torch.Tensor:
    """a 0-dim fp32 scalar (the reference's CPU scalar tensors)."""
    return torch.as_tensor(v, dtype=F).reshape(())


# ----------------------------------------------------------------------------
# CFG mix
# ----------------------------------------------------------------------------
def cfg_mix(eps_uc: torch.Tensor, eps_c: torch.Tensor, lam: float) -> torch.Tensor:
    """eps_hat = eps_uc + lam * (eps_c - eps_uc), every op rounded to eps' dtype.

    With fp16 eps each of the three ops rounds to fp16; the python scalar ``lam``
    enters the multiply as an fp32 value (opmath), not as an fp16 one.
    """
    if eps_uc.dtype == H:
        d = _h(_f(eps_c) - _f(eps_uc))
        e = _h(_f(d) * _s(lam))
        return _h(_f(eps_uc) + _f(e))
    d = eps_c - eps_uc
    return eps_uc + d * _s(lam)


def _smul_first(s, x: torch.Tensor, semantics: str = "cpu") -> torch.Tensor:
    """``s * x`` with the 0-dim fp32 TENSOR scalar written FIRST, as the reference
    writes ``(1-at).sqrt() * noise_pred`` or ``-torch.exp(-h) * uncond_denoised``.

    torch-CPU semantics (probed on the torch in this image, and what the golden
    vectors contain): with an fp16 tensor the scalar operand is first cast to the
    common dtype, i.e. ROUNDED TO FP16; the product is then formed in fp32 and
    rounded once.  (torch-CUDA keeps the scalar in fp32 here: the HIP kernel takes
    fp32 coefficients and the host chooses whether to pre-round them - see
    ``cfgpp_amd.coeffs``.)  ``semantics="cuda"`` restates that GPU behaviour: the scalar enters the product as
    an fp32 opmath value.  It is what the product runs with by default, and it is NOT pinned by golden vectors
    (they can only be recorded on torch-CPU here).  Result keeps x's dtype."""
    s = _s(s)
    if x.dtype == H:
        return _h(_f(x) * (_f(_h(s)) if semantics == "cpu" else s))
    return x * s


def _scale_eps(coef: torch.Tensor, eps: torch.Tensor, semantics: str = "cpu") -> torch.Tensor:
    """``coef * eps`` (scalar first), returned as fp32 for the fp32 latent update."""
    return _f(_smul_first(coef, eps, semantics))


# ----------------------------------------------------------------------------
# DDIM family: forward / inversion x CFG / CFG++
# ----------------------------------------------------------------------------
def ddim_coeffs(a_tweedie, a_renoise):
    """(c1, c2, c3, c4) = sqrt(1-a_tw), sqrt(a_tw), sqrt(a_rn), sqrt(1-a_rn), fp32
    scalar ops exactly as ``(1-at).sqrt()`` / ``at.sqrt()`` on 0-dim fp32 tensors."""
    a_tw, a_rn = _s(a_tweedie), _s(a_renoise)
    return (1 - a_tw).sqrt(), a_tw.sqrt(), a_rn.sqrt(), (1 - a_rn).sqrt()


def ddim_step(z, eps_uc, eps_c, lam, a_tweedie, a_renoise, tweedie_uc: bool, renoise_uc: bool, sqrt4=None,
              semantics: str = "cpu"):
    """One generalised DDIM update.

        z0t = (z - sqrt(1-a_tw) * A) / sqrt(a_tw)
        z'  = sqrt(a_rn) * z0t + sqrt(1-a_rn) * B

    forward CFG    : a_tw=alpha(t),      a_rn=alpha(t-skip), A=eps_hat, B=eps_hat
    forward CFG++  : a_tw=alpha(t),      a_rn=alpha(t-skip), A=eps_hat, B=eps_uc
    inversion CFG  : a_tw=alpha(t-skip), a_rn=alpha(t),      A=eps_hat, B=eps_hat
    inversion CFG++: a_tw=alpha(t-skip), a_rn=alpha(t),      A=eps_uc,  B=eps_hat

    z fp32 (text-to-image: ``torch.randn``): z0t and z' are fp32 (fp16 eps products are rounded to fp16 first).
    z fp16 (inversion / edit: the latent is the fp16 ``vae.encode`` sample, latent_diffusion.py:168,527-541;
    latent_sdxl.py:307,989-1011): every op rounds to fp16 -
        pa = h(c1*A); z0t = h(h(z - pa) / c2); z' = h(h(c3*z0t) + h(c4*B))
    with the scalar-first products (c1, c3, c4) following ``semantics`` and the divisor c2 (scalar second)
    fp32 on every backend (probed on torch-CPU; pinned by tests/golden/sampler_golden_h16.npz).
    ``sqrt4`` = (c1, c2, c3, c4) overrides the torch ``sqrt`` evaluation with pinned values
    (``torch.sqrt`` differs by 1 ulp between hosts; see cfgpp_amd/schedule.py).
    """
    if sqrt4 is not None:
        c1, c2, c3, c4 = (_s(v) for v in sqrt4)
    else:
        c1, c2, c3, c4 = ddim_coeffs(a_tweedie, a_renoise)
    eps_hat = cfg_mix(eps_uc, eps_c, lam)
    A = eps_uc if tweedie_uc else eps_hat
    B = eps_uc if renoise_uc else eps_hat
    if z.dtype == H:
        pa = _smul_first(c1, A, semantics)
        z0t = _div_s(_sub(z, pa), c2, semantics)
        zn = _add(_smul_first(c3, z0t, semantics), _smul_first(c4, B, semantics))
        return z0t, zn
    zf = _f(z)
    z0t = _div_s(zf - _scale_eps(c1, A, semantics), c2, semantics)
    zn = c3 * z0t + _scale_eps(c4, B, semantics)
    return z0t, zn


# ----------------------------------------------------------------------------
# k-diffusion family (latent x is fp16 in the reference)
# ----------------------------------------------------------------------------
def kdiff_input_div(x, sigma):
    """SD1.5 ``calculate_input``: x / (sigma**2 + 1)**0.5 (latent_diffusion.py:229)."""
    s = (_s(sigma) ** 2 + 1) ** 0.5
    if x.dtype == H:
        return _h(_f(x) / s)
    return x / s


def kdiff_input_mul(x, c_in):
    """SDXL 2M: ``x * c_in`` (latent_sdxl.py:901)."""
    if x.dtype == H:
        return _h(_f(x) * _s(c_in))
    return x * _s(c_in)


def _mul_s(x, s):
    """``x * s`` with the tensor written FIRST (or a python scalar in either
    position): the scalar stays fp32 (opmath); result in x's dtype."""
    if x.dtype == H:
        return _h(_f(x) * _s(s))
    return x * _s(s)


def _div_s(x, s, semantics: str = "cpu"):
    """``x / s`` with a 0-dim CPU scalar (or python number) divisor.  torch-CPU: IEEE division.  torch on a GPU
    (``semantics="cuda"``): the reciprocal of a CPU-scalar divisor is taken once on the host in fp32 and the kernel
    multiplies (ATen div_true_kernel_cuda) - pinned on the GPU box by tests/test_gpu_torch_semantics.py."""
    s = _s(s)
    if semantics == "cuda":
        inv = _s(1.0) / s
        return _h(_f(x) * inv) if x.dtype == H else x * inv
    if x.dtype == H:
        return _h(_f(x) / s)
    return x / s


def _add(a, b):
    if a.dtype == H and b.dtype == H:
        return _h(_f(a) + _f(b))
    return _f(a) + _f(b)


def _sub(a, b):
    if a.dtype == H and b.dtype == H:
        return _h(_f(a) - _f(b))
    return _f(a) - _f(b)


def kdiff_denoised(x, eps_uc, eps_c, lam, sigma, xl_form: bool = False, semantics: str = "cpu"):
    """SD1.5 (latent_diffusion.py:232-241): ``denoised = x - eps_hat*sigma``,
    ``uncond_denoised = x - eps_uc*sigma`` (tensor first -> fp32 sigma).
    SDXL 2M (latent_sdxl.py:895-906, ``xl_form``): ``x + c_out*eps`` with
    ``c_out = -sigma`` written first (-> fp16-rounded scalar)."""
    eps_hat = cfg_mix(eps_uc, eps_c, lam)
    if xl_form:
        c_out = -_s(sigma)
        den = _add(x, _smul_first(c_out, eps_hat, semantics))
        uden = _add(x, _smul_first(c_out, eps_uc, semantics))
    else:
        den = _sub(x, _mul_s(eps_hat, sigma))
        uden = _sub(x, _mul_s(eps_uc, sigma))
    return den, uden


def euler_step(x, den, d_from, sigma, sigma_next, semantics: str = "cpu"):
    """x' = den + ((x - d_from)/sigma.item()) * sigma_next
    (latent_diffusion.py:708-710; ``d_from`` = uncond_denoised for CFG++, den for CFG).  ``/ sigma.item()`` is a division by
    a python number: IEEE on torch-CPU, a multiply by the host-side fp32 reciprocal on a GPU (``semantics``, see _div_s)."""
    d = _div_s(_sub(x, d_from), float(sigma), semantics)
    return _add(den, _mul_s(d, sigma_next))


def dpm2m_coeffs(sigmas, i):
    """h, r, exp(-h), expm1(-h) as fp32 0-dim tensors, mirroring the reference's
    ``t_fn = -log(sigma)`` arithmetic (latent_diffusion.py:856-862)."""
    t_fn = lambda s: _s(s).log().neg()
    t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
    h = t_next - t
    out = dict(h=h, exp_mh=torch.exp(-h), expm1_mh=(-h).expm1())
    if i > 0:
        h_last = t - t_fn(sigmas[i - 1])
        out["r"] = h_last / h
    return out


def dpm2m_step(x, den, uden, old, sigmas, i, variant: str, semantics: str = "cpu"):
    """One DPM-Solver++(2M) update on an fp16 (or fp32) latent.

    variant "cfg"      : latent_diffusion.py:482-490  (extra uses den, old=den)
    variant "cfgpp_sd" : latent_diffusion.py:858-866  (first branch uses uden in to_d;
                         extra1 = -e^{-h}*uden - expm1(-h)*(den - old)/(2r); old=uden)
    variant "cfgpp_xl" : latent_sdxl.py:911-919       (… (uden - old)/(2r); old=uden)
    Returns (x_next, new_old).
    """
    sig, sig_next = _s(sigmas[i]), _s(sigmas[i + 1])
    first = old is None or float(sig_next) == 0.0
    d_from = den if variant == "cfg" else uden
    if first:
        xn = euler_step(x, den, d_from, sig, sig_next, semantics)
    else:
        c = dpm2m_coeffs(sigmas, i)
        lead = den if variant == "cfg" else uden           # multiplied by -exp(-h)
        diff_a = uden if variant == "cfgpp_xl" else den     # (diff_a - old)
        # extra1 = -exp(-h)*lead - expm1(-h) * (diff_a - old) / (2r)
        term1 = _smul_first(-c["exp_mh"], lead, semantics)
        term2 = _div_s(_smul_first(c["expm1_mh"], _sub(diff_a, old), semantics), 2 * c["r"], semantics)
        extra1 = _sub(term1, term2)
        extra2 = _smul_first(c["exp_mh"], x, semantics)
        xn = _add(_add(den, extra1), extra2)
    new_old = den if variant == "cfg" else uden
    return xn, new_old


# ----------------------------------------------------------------------------
# whole-loop drivers (used by tests and by bench.py's cpu_baseline leg)
# ----------------------------------------------------------------------------
def sample_ddim(unet_fn, zT, tables, lam, cfgpp=True, wrap_index=False, callback_fn=None, semantics: str = "cpu"):
    """DDIM / DDIM-CFG++ forward loop on fp32 latents.

    unet_fn(z, t) -> (eps_uc, eps_c).  ``wrap_index`` selects the SDXL unguarded
    index rule (quirk Q3).  Returns (z0t, zt, trajectory-less)."""
    zt = zT.clone() if zT.dtype == H else zT.clone().to(F)
    z0t = None
    ts = tables.timesteps
    ts = ts.int() if wrap_index else ts
    for step, t in enumerate(ts):
        if wrap_index:
            at, at_prev = tables.alpha_wrap(t), tables.alpha_wrap(int(t) - tables.skip)
        else:
            at, at_prev = tables.alpha(t), tables.alpha(int(t) - tables.skip)
        eps_uc, eps_c = unet_fn(zt, t)
        z0t, zt = ddim_step(zt, eps_uc, eps_c, lam, at, at_prev, tweedie_uc=False, renoise_uc=cfgpp,
                            sqrt4=tables.ddim_sqrt_coeffs(t, wrap=wrap_index), semantics=semantics)
        if callback_fn is not None:
            kw = callback_fn(step, t, {"z0t": z0t, "zt": zt, "decode": None})
            z0t, zt = kw["z0t"], kw["zt"]
    return z0t, zt


def invert_ddim(unet_fn, z0, tables, lam, cfgpp=True, semantics: str = "cpu"):
    """DDIM inversion loop (latent_diffusion.py:888-910 / :160-182)."""
    zt = z0.clone() if z0.dtype == H else z0.clone().to(F)
    for t in reversed(tables.timesteps):
        at, at_prev = tables.alpha(t), tables.alpha(int(t) - tables.skip)
        eps_uc, eps_c = unet_fn(zt, t)
        _, zt = ddim_step(zt, eps_uc, eps_c, lam, at_prev, at, tweedie_uc=cfgpp, renoise_uc=False,
                          sqrt4=tables.ddim_sqrt_coeffs(t, inversion=True), semantics=semantics)
    return zt
