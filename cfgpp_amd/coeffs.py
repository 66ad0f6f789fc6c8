"""Host-side coefficient tables for the fused step kernels.

The reference computes every per-step scalar as a 0-dim fp32 CPU tensor
(``at = alphas_cumprod[t]``, ``(1-at).sqrt()``, ``torch.exp(-h)`` ...) and lets
PyTorch type promotion combine it with fp16 eps / fp32 latents.  The HIP
kernels take plain fp32 coefficients; this module evaluates them with the same
torch scalar expressions (bit-identical fp32) and applies the one promotion
rule that is not visible in the formulas:

  a 0-dim fp32 *tensor* written FIRST in ``s * x`` with x fp16 is rounded to
  fp16 before the multiply on the torch-CPU path the golden vectors were
  recorded on (``scalar_semantics="cpu"``);  torch-CUDA - what the reference
  actually runs on - passes a CPU 0-dim scalar to the kernel as an fp32
  opmath value (``scalar_semantics="cuda"``, the default of the HIP engine).

The second backend-dependent rule is division by a scalar: ``x / s`` with ``s`` a CPU 0-dim tensor or a python
number is an IEEE division on torch-CPU, but torch's GPU ``div`` kernel takes the reciprocal of a CPU scalar
divisor once on the host (fp32) and multiplies (ATen ``div_true_kernel_cuda``, ``iter.is_cpu_scalar(2)``).
:func:`divisor` encodes the choice in the sign the kernels understand (include/cfgpp.h).  Both ``"cuda"`` rules are
pinned on the GPU box against torch-ROCm evaluating the reference's own expressions
(tests/test_gpu_torch_semantics.py).

Reference lines: latent_diffusion.py:655-666, 901-908, 849-866; latent_sdxl.py:732-744, 892-919.
"""
from __future__ import annotations

import torch

F32 = torch.float32


def _s(v) -> torch.Tensor:
    return torch.as_tensor(v, dtype=F32).reshape(())


def _first(s: torch.Tensor, half_operand: bool, semantics: str) -> float:
    """value of a scalar written first in ``s * fp16_tensor``."""
    if half_operand and semantics == "cpu":
        return float(s.to(torch.float16).to(F32))
    return float(s)


def divisor(s, semantics: str) -> float:
    """the value the kernels take for ``x / s``: ``s`` itself (IEEE division, torch-CPU) or ``-fl32(1/s)`` (the kernel
    multiplies by the host-side fp32 reciprocal: torch's GPU ``div`` with a CPU-scalar divisor)."""
    s = _s(s)
    if semantics == "cuda":
        return -float(_s(1.0) / s)
    return float(s)


def ddim_coeffs(a_tweedie, a_renoise, eps_half: bool = True, semantics: str = "cpu"):
    """(c1, c2, c3, c4) for cfgpp_step_ddim:
    z0t = (z - c1*A)/c2 ; z' = c3*z0t + c4*B with c1 = sqrt(1-a_tw), c2 = sqrt(a_tw),
    c3 = sqrt(a_rn), c4 = sqrt(1-a_rn)  (latent_diffusion.py:663,666)."""
    a_tw, a_rn = _s(a_tweedie), _s(a_renoise)
    c1, c2, c3, c4 = (1 - a_tw).sqrt(), a_tw.sqrt(), a_rn.sqrt(), (1 - a_rn).sqrt()
    return (_first(c1, eps_half, semantics), divisor(c2, semantics), float(c3), _first(c4, eps_half, semantics))


def ddim_coeffs_pinned(sqrt4, eps_half: bool = True, semantics: str = "cpu", z_half: bool = False, device_alpha=None):
    """Same as :func:`ddim_coeffs` but from the pinned sqrt tables
    (``SchedulerTables.ddim_sqrt_coeffs``) - bit-stable across hosts.  ``z_half``: the latent itself is
    fp16 (inversion / edit paths), so ``at_prev.sqrt() * z0t`` is another scalar-first product with an
    fp16 tensor (c3); the divisor ``/ at.sqrt()`` (c2, scalar second) stays fp32 on every backend and follows
    :func:`divisor`.

    ``device_alpha`` ("tw" / "rn" / None): that alpha is ``final_alpha_cumprod.to(device)`` (latent_diffusion.py:80,
    88-90: the ``t - skip < 0`` step of the SD1.5 loops), a 0-dim DEVICE tensor.  On a GPU it is then an ordinary
    operand, not a CPU scalar: it is cast to the common dtype of the op (fp16 whenever the tensor operand is fp16) and a
    division by it is a true division.  Only matters under ``semantics="cuda"`` (on torch-CPU every scalar is a CPU
    tensor)."""
    c1, c2, c3, c4 = (_s(v) for v in sqrt4)
    if semantics == "cuda" and device_alpha == "rn":
        return (_first(c1, eps_half, semantics), divisor(c2, semantics), _first(c3, z_half, "cpu"), _first(c4, eps_half, "cpu"))
    if semantics == "cuda" and device_alpha == "tw":
        return (_first(c1, eps_half, "cpu"), _first(c2, z_half, "cpu"), _first(c3, z_half, semantics), _first(c4, eps_half, semantics))
    return (_first(c1, eps_half, semantics), divisor(c2, semantics), _first(c3, z_half, semantics), _first(c4, eps_half, semantics))


def kdiff_input_scale_sd(sigma, semantics: str = "cpu") -> float:
    """divisor of ``x / (sigma**2 + 1)**0.5`` (latent_diffusion.py:229-230), as :func:`divisor` encodes it."""
    return divisor((_s(sigma) ** 2 + 1) ** 0.5, semantics)


def kdiff_coeffs(lam, sigmas, i, first: bool, xl_form: bool, semantics: str = "cpu"):
    """coef[9] for cfgpp_step_kdiff at step i and whether the Euler branch is taken.

    ``first`` = ``old_denoised is None``.  Mirrors latent_diffusion.py:856-865 /
    latent_sdxl.py:909-918 scalar arithmetic (t_fn = -log sigma)."""
    sig, sig_next = _s(sigmas[i]), _s(sigmas[i + 1])
    euler = bool(first or float(sig_next) == 0.0)
    t_fn = lambda s: s.log().neg()  # noqa: E731
    coef = [float(lam), float(sig), _first(-sig, True, semantics), divisor(sig.item(), semantics), float(sig_next), 0.0, 0.0, 1.0, 0.0]
    if not euler:
        t, t_next = t_fn(sig), t_fn(sig_next)
        h = t_next - t
        h_last = t - t_fn(_s(sigmas[i - 1]))
        r = h_last / h
        coef[5] = _first(-torch.exp(-h), True, semantics)
        coef[6] = _first((-h).expm1(), True, semantics)
        coef[7] = divisor(2 * r, semantics)
        coef[8] = _first(torch.exp(-h), True, semantics)
    return coef, euler
