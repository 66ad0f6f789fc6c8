"""-m gpu: the product's default scalar semantics ("cuda", cfgpp_amd/coeffs.py) pinned by OBSERVATION.

The golden vectors are torch-CPU recordings; the reference itself runs on a GPU, where two promotion rules differ
(a CPU 0-dim fp32 scalar written first in ``s * fp16_tensor`` stays fp32; ``x / cpu_scalar`` is ``x * (1/scalar)``).
The GPU box has torch-ROCm - the same ATen kernels, hipified - so here torch itself evaluates the reference's own
expressions, written as the reference writes them (CPU 0-dim fp32 scalars taken from CPU tables, device tensors, python
``cfg_guidance``, under ``torch.autocast('cuda', float16)`` like ``sample()``), and the HIP step kernels driven by
``coeffs.*(semantics="cuda")`` must reproduce every element BIT FOR BIT.

    DDIM / DDIM-CFG++ forward     latent_diffusion.py:655-666, 280-286      latent_sdxl.py:738-744
    DDIM inversion CFG / CFG++    latent_diffusion.py:175-180, 901-908      latent_sdxl.py:315-318, 970-973
    k-diffusion input / denoised  latent_diffusion.py:229-241               latent_sdxl.py:895-906
    Euler / DPM++2M CFG / CFG++   latent_diffusion.py:329-333, 477-490, 706-710, 853-866   latent_sdxl.py:909-919
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd import engine
    return engine


def _rand(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


SHAPE = (2, 4, 64, 64)


@pytest.mark.parametrize("z_half", [False, True])
@pytest.mark.parametrize("mode", ["cfgpp", "cfg", "inv_cfgpp", "inv_cfg"])
def test_ddim_step_equals_torch_rocm_on_the_references_expressions(E, mode, z_half):
    from cfgpp_amd.coeffs import ddim_coeffs
    from cfgpp_amd.schedule import SchedulerTables
    tb = SchedulerTables(50)
    alphas = tb.alphas_cumprod                   # the shifted table cat([1.0], abar) of latent_diffusion.py:81 (a CPU tensor)
    lam = 0.6 if "cfgpp" in mode else 7.5
    bad, detail = 0, []
    for k, t in enumerate([981, 501, 21]):
        at, at_prev = alphas[t], alphas[t - 20]                            # 0-dim fp32 CPU tensors
        zt = _rand(SHAPE, 10 + k, 1.3, torch.float16 if z_half else torch.float32).cuda()
        noise_uc, noise_c = (_rand(SHAPE, 20 + 2 * k + j, 1.0, torch.float16).cuda() for j in range(2))
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            noise_pred = noise_uc + lam * (noise_c - noise_uc)
            if mode == "cfgpp":
                z0t = (zt - (1 - at).sqrt() * noise_pred) / at.sqrt()
                zn = at_prev.sqrt() * z0t + (1 - at_prev).sqrt() * noise_uc
                co, tw, rn = ddim_coeffs(at, at_prev, True, "cuda"), False, True
            elif mode == "cfg":
                z0t = (zt - (1 - at).sqrt() * noise_pred) / at.sqrt()
                zn = at_prev.sqrt() * z0t + (1 - at_prev).sqrt() * noise_pred
                co, tw, rn = ddim_coeffs(at, at_prev, True, "cuda"), False, False
            elif mode == "inv_cfgpp":
                z0t = (zt - (1 - at_prev).sqrt() * noise_uc) / at_prev.sqrt()
                zn = at.sqrt() * z0t + (1 - at).sqrt() * noise_pred
                co, tw, rn = ddim_coeffs(at_prev, at, True, "cuda"), True, False
            else:
                z0t = (zt - (1 - at_prev).sqrt() * noise_pred) / at_prev.sqrt()
                zn = at.sqrt() * z0t + (1 - at).sqrt() * noise_pred
                co, tw, rn = ddim_coeffs(at_prev, at, True, "cuda"), False, False
        assert z0t.dtype == zt.dtype and zn.dtype == zt.dtype
        zk, z0k = zt.clone(), torch.empty_like(zt)
        E.step_ddim(zk, z0k, noise_uc, noise_c, lam, co, tw, rn)
        torch.cuda.synchronize()
        nb = (int((z0k != z0t).sum()), int((zk != zn).sum()))
        detail.append((t, nb))
        bad += sum(nb)
    assert bad == 0, f"{mode} z_half={z_half}: elements differing from torch-ROCm per step (t, (z0t, zt)): {detail}"


@pytest.mark.parametrize("variant,xl_form,lam", [(1, False, 0.6), (0, False, 7.5), (2, True, 0.6)])
def test_kdiff_steps_equal_torch_rocm_on_the_references_expressions(E, variant, xl_form, lam):
    """Euler branch (first step) and DPM++2M branch of dpm++_2m(_cfg++) in the SD1.5 form (x - eps*sigma, x / sqrt(sigma^2+1))
    and the SDXL form (x + c_out*eps, x * c_in)"""
    from cfgpp_amd.coeffs import kdiff_coeffs, kdiff_input_scale_sd
    from cfgpp_amd.schedule import SchedulerTables
    tb = SchedulerTables(20)
    sigmas = tb.karras_sigmas()                                            # CPU fp32, like get_sigmas_karras(..., device='cpu')
    t_fn = lambda sigma: sigma.log().neg()  # noqa: E731
    bad, detail = 0, []
    for i in (0, 1, 7, 18):
        sigma = sigmas[i]
        x = _rand(SHAPE, 30 + i, float(sigma), torch.float16).cuda()
        old_denoised = None if i == 0 else _rand(SHAPE, 31 + i, 1.0, torch.float16).cuda()
        noise_uc, noise_c = (_rand(SHAPE, 40 + 2 * i + j, 1.0, torch.float16).cuda() for j in range(2))
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            # UNet input scaling
            if xl_form:
                at = (1 / (sigma ** 2 + 1)).to(torch.float32)              # any CPU 0-dim value does: c_in = at.clone().sqrt()
                c_in = at.clone().sqrt()
                xc = x * c_in
            else:
                xc = x / (sigma ** 2 + 1) ** 0.5
            noise_pred = noise_uc + lam * (noise_c - noise_uc)
            if xl_form:
                c_out = -sigma.clone()
                denoised = x + c_out * noise_pred
                uncond_denoised = x + c_out * noise_uc
            else:
                denoised = x - noise_pred * sigma
                uncond_denoised = x - noise_uc * sigma
            d_from = denoised if variant == 0 else uncond_denoised
            t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
            h = t_next - t
            if old_denoised is None or sigmas[i + 1] == 0:
                xn = denoised + ((x - d_from) / sigmas[i].item()) * sigmas[i + 1]
            else:
                h_last = t - t_fn(sigmas[i - 1])
                r = h_last / h
                lead = d_from
                diff_a = uncond_denoised if variant == 2 else denoised
                extra1 = -torch.exp(-h) * lead - (-h).expm1() * (diff_a - old_denoised) / (2 * r)
                extra2 = torch.exp(-h) * x
                xn = denoised + extra1 + extra2
        assert xc.dtype == torch.float16 and xn.dtype == torch.float16
        xck = torch.empty_like(x)
        if xl_form:
            E.kdiff_input(x, xck, float(c_in), 1)
        else:
            E.kdiff_input(x, xck, kdiff_input_scale_sd(sigma, "cuda"), 0)
        coef, euler = kdiff_coeffs(lam, sigmas, i, old_denoised is None, xl_form=xl_form, semantics="cuda")
        xk, denk = x.clone(), torch.empty_like(x)
        oldk = old_denoised.clone() if old_denoised is not None else torch.empty_like(x)
        E.step_kdiff(xk, denk, oldk, noise_uc, noise_c, coef, variant, xl_form, euler, True)
        torch.cuda.synchronize()
        new_old = denoised if variant == 0 else uncond_denoised
        nb = (int((xck != xc).sum()), int((denk != denoised).sum()), int((xk != xn).sum()), int((oldk != new_old).sum()))
        detail.append((i, nb))
        bad += sum(nb)
    assert bad == 0, f"variant {variant} xl_form={xl_form}: elements differing from torch-ROCm per step (i, (xc, denoised, x, old)): {detail}"


@pytest.mark.parametrize("cfgpp", [False, True])
def test_ancestral_euler_step_equals_torch_rocm_on_the_references_expressions(E, cfgpp):
    """euler_a / euler_a_cfg++ (latent_diffusion.py:349-390, 726-766): ``to_d``'s ``/ sigma.item()`` is a python-number divisor
    (reciprocal multiply on the GPU) and ``d * sigma_down`` a tensor-first product; the coefficients are the ones
    ``StableDiffusion._ancestral_loop`` hands to cfgpp_step_kdiff."""
    from cfgpp_amd import coeffs as K
    from cfgpp_amd.schedule import get_ancestral_step
    from cfgpp_amd.schedule import SchedulerTables
    tb = SchedulerTables(20)
    sigmas = tb.karras_sigmas()
    lam = 0.6 if cfgpp else 7.5
    bad, detail = 0, []
    for i in (0, 3, 11, 18, 19):
        sigma = sigmas[i]
        sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1])
        x = _rand(SHAPE, 60 + i, float(sigma), torch.float16).cuda()
        noise_uc, noise_c = (_rand(SHAPE, 70 + 2 * i + j, 1.0, torch.float16).cuda() for j in range(2))
        with torch.autocast(device_type="cuda", dtype=torch.float16):
            noise_pred = noise_uc + lam * (noise_c - noise_uc)
            denoised = x - noise_pred * sigma
            uncond_denoised = x - noise_uc * sigma
            d = (x - (uncond_denoised if cfgpp else denoised)) / sigma.item()
            xn = denoised + d * sigma_down
        assert xn.dtype == torch.float16
        coef = [float(lam), float(sigma), 0.0, K.divisor(sigma.item(), "cuda"), float(sigma_down), 0.0, 0.0, 1.0, 0.0]
        xk, denk = x.clone(), torch.empty_like(x)
        E.step_kdiff(xk, denk, None, noise_uc, noise_c, coef, 1 if cfgpp else 0, False, True, False)
        torch.cuda.synchronize()
        nb = (int((denk != denoised).sum()), int((xk != xn).sum()))
        detail.append((i, nb))
        bad += sum(nb)
    assert bad == 0, f"cfgpp={cfgpp}: elements differing from torch-ROCm per step (i, (denoised, x)): {detail}"


@pytest.mark.parametrize("z_half", [False, True])
@pytest.mark.parametrize("flow", ["forward_last_step", "inversion_first_step"])
def test_device_resident_final_alpha(E, flow, z_half):
    """``final_alpha_cumprod.to(device)`` (latent_diffusion.py:80, latent_sdxl.py:66) is a 0-dim DEVICE tensor: on the
    ``t - skip < 0`` step its products / quotient follow the ordinary-operand rule (cast to the common dtype, true
    division), not the CPU-scalar rule - ``coeffs.ddim_coeffs_pinned(device_alpha=...)``."""
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    tb = SchedulerTables(50)
    alphas = tb.alphas_cumprod                   # already the shifted table cat([1.0], abar)
    t = 1
    at = alphas[t]
    at_prev = tb.final_alpha_cumprod.clone().cuda()          # alpha(t - skip) for t - skip < 0
    zt = _rand(SHAPE, 3, 1.0, torch.float16 if z_half else torch.float32).cuda()
    noise_uc, noise_c = (_rand(SHAPE, 4 + j, 1.0, torch.float16).cuda() for j in range(2))
    lam = 0.6
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        noise_pred = noise_uc + lam * (noise_c - noise_uc)
        if flow == "forward_last_step":
            z0t = (zt - (1 - at).sqrt() * noise_pred) / at.sqrt()
            zn = at_prev.sqrt() * z0t + (1 - at_prev).sqrt() * noise_uc
            co = ddim_coeffs_pinned(tb.ddim_sqrt_coeffs(t), True, "cuda", z_half=z_half, device_alpha="rn")
            tw, rn = False, True
        else:
            z0t = (zt - (1 - at_prev).sqrt() * noise_uc) / at_prev.sqrt()
            zn = at.sqrt() * z0t + (1 - at).sqrt() * noise_pred
            co = ddim_coeffs_pinned(tb.ddim_sqrt_coeffs(t, inversion=True), True, "cuda", z_half=z_half, device_alpha="tw")
            tw, rn = True, False
    zk, z0k = zt.clone(), torch.empty_like(zt)
    E.step_ddim(zk, z0k, noise_uc, noise_c, lam, co, tw, rn)
    torch.cuda.synchronize()
    bad = int((z0k != z0t).sum()) + int((zk != zn).sum())
    assert bad == 0, f"{flow} z_half={z_half}: {bad} elements differ from torch-ROCm"
