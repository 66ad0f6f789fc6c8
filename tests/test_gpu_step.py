"""-m gpu: fused step kernels vs the reference's golden vectors and vs the CPU oracle, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd import engine
    return engine


CASES = [("G2/sd_ddim_cfgpp_h", 50, "ddim", 0.6, True, False), ("G2/sd_ddim_cfgpp_f", 50, "ddim", 0.6, True, False),
         ("G2/sd_ddim_cfg_h", 10, "ddim", 7.5, False, False), ("G3/sd_inv_cfgpp", 10, "ddim", 0.6, True, False),
         ("G3/sd_inv_cfg", 10, "ddim", 2.0, False, False), ("G2/xl_ddim_cfgpp", 50, "ddim", 0.6, True, True),
         ("G2/xl_light_ddim_cfgpp", 4, "lightning", 1.0, True, True), ("G3/xl_edit_cfgpp", 10, "ddim", 0.6, True, False)]


@pytest.mark.parametrize("tag,nfe,kind,lam,cfgpp,wrap", CASES)
def test_ddim_family_vs_golden(E, golden, tag, nfe, kind, lam, cfgpp, wrap):
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    g, _ = golden
    tb = SchedulerTables(nfe, kind)
    z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
    off = z.shape[0] - z0.shape[0]
    half = e.dtype == torch.float16
    ts = tb.timesteps.int() if wrap else tb.timesteps

    def run(i, co, tw, rn):
        zi = z[i][0:1].contiguous().cuda()
        z0o = torch.empty_like(zi)
        E.step_ddim(zi, z0o, e[i][0:1].contiguous().cuda(), e[i][1:2].contiguous().cuda(), lam, co, tw, rn)
        return z0o.cpu(), zi.cpu()
    for i, t in enumerate(ts):
        a, b = run(off + i, ddim_coeffs_pinned(tb.ddim_sqrt_coeffs(t, wrap=wrap), eps_half=half), False, cfgpp)
        assert torch.equal(a, z0[i]) and torch.equal(b, zt[i]), f"{tag} step {i}"
    for i, t in enumerate(reversed(tb.timesteps)):
        if i >= off:
            break
        _, b = run(i, ddim_coeffs_pinned(tb.ddim_sqrt_coeffs(t, inversion=True), eps_half=half), cfgpp, False)
        assert torch.equal(b, z[i + 1][0:1]), f"{tag} inversion step {i}"


@pytest.mark.parametrize("variant,xl_form,solver", [(0, False, "dpm2m"), (1, False, "dpm2m"), (2, True, "dpm2m"), (1, False, "euler"), (0, False, "euler")])
def test_kdiff_kernel_vs_emulation_large(E, variant, xl_form, solver):
    """k-diffusion step kernel == its CPU emulation (tests/mock_engine.py, itself pinned to the golden
    trajectories in test_solver_cpu.py) on a full-size fp16 latent batch, incl. the 2M branch."""
    from cfgpp_amd.coeffs import kdiff_coeffs
    from cfgpp_amd.schedule import SchedulerTables
    from mock_engine import emulate_kdiff_input, emulate_step_kdiff
    tb = SchedulerTables(20)
    sig = tb.karras_sigmas()
    g = torch.Generator().manual_seed(3)
    n = (8, 4, 64, 64)
    x = (torch.randn(n, generator=g) * 3).half()
    old = torch.randn(n, generator=g).half()
    for i in (0, 5, 18, 19):
        eu, ec = torch.randn(n, generator=g).half(), torch.randn(n, generator=g).half()
        first = (solver == "euler") or i == 0
        coef, euler = kdiff_coeffs(0.6, sig, i, first, xl_form)
        xr, dr, orr = x.clone(), torch.empty_like(x), old.clone()
        emulate_step_kdiff(xr, dr, orr, eu, ec, coef, variant, xl_form, euler, solver != "euler")
        xd, dd, od = x.clone().cuda(), torch.empty_like(x).cuda(), old.clone().cuda()
        E.step_kdiff(xd, dd, od, eu.cuda(), ec.cuda(), coef, variant, xl_form, euler, solver != "euler")
        assert torch.equal(xd.cpu(), xr) and torch.equal(dd.cpu(), dr) and torch.equal(od.cpu(), orr), f"step {i}"
        xc_r, xc_d = torch.empty_like(x), torch.empty_like(x).cuda()
        emulate_kdiff_input(x, xc_r, 1.7321, 0 if not xl_form else 1)
        E.kdiff_input(x.cuda(), xc_d, 1.7321, 0 if not xl_form else 1)
        assert torch.equal(xc_d.cpu(), xc_r)


def test_ddim_step_full_size_properties(E):
    """BASELINE sizes ([8,4,64,64] and [16,4,128,128]): kernel == oracle on random data, and lambda = 0 with
    identical eps reduces CFG and CFG++ to the same update (size-independent property)."""
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    from oracle import sampler as O
    tb = SchedulerTables(50)
    g = torch.Generator().manual_seed(5)
    for shape in ((8, 4, 64, 64), (16, 4, 128, 128)):
        z = torch.randn(shape, generator=g)
        eu, ec = torch.randn(shape, generator=g).half(), torch.randn(shape, generator=g).half()
        t = tb.timesteps[7]
        s4 = tb.ddim_sqrt_coeffs(t)
        a, b = O.ddim_step(z, eu, ec, 0.6, None, None, False, True, sqrt4=s4)
        zd, z0d = z.clone().cuda(), torch.empty_like(z).cuda()
        E.step_ddim(zd, z0d, eu.cuda(), ec.cuda(), 0.6, ddim_coeffs_pinned(s4), False, True)
        assert torch.equal(z0d.cpu(), a) and torch.equal(zd.cpu(), b)
        z1, z2 = z.clone().cuda(), z.clone().cuda()
        o1, o2 = torch.empty_like(z1), torch.empty_like(z2)
        E.step_ddim(z1, o1, eu.cuda(), eu.cuda(), 0.0, ddim_coeffs_pinned(s4), False, True)
        E.step_ddim(z2, o2, eu.cuda(), eu.cuda(), 0.0, ddim_coeffs_pinned(s4), False, False)
        assert torch.equal(z1, z2) and torch.equal(o1, o2)


def test_denoise_and_lincomb_kernels_vs_emulation(E):
    """building blocks of the ancestral / 2-stage samplers: bit-exact vs the CPU emulation that reproduces the
    reference's golden trajectories (tests/test_solver_cpu.py::test_sd_ancestral_trajectory)."""
    from mock_engine import emulate_kdiff_denoise, emulate_lincomb
    g = torch.Generator().manual_seed(9)
    n = (8, 4, 64, 64)
    x, y, z = ((torch.randn(n, generator=g) * 2).half() for _ in range(3))
    eu, ec = torch.randn(n, generator=g).half(), torch.randn(n, generator=g).half()
    dr, ur = torch.empty_like(x), torch.empty_like(x)
    emulate_kdiff_denoise(x, eu, ec, 0.6, 3.217, dr, ur)
    dd, ud = torch.empty_like(x).cuda(), torch.empty_like(x).cuda()
    E.kdiff_denoise(x.cuda(), eu.cuda(), ec.cuda(), 0.6, 3.217, dd, ud)
    assert torch.equal(dd.cpu(), dr) and torch.equal(ud.cpu(), ur)
    for mode, (a, b) in ((0, (0.7312, -0.2466)), (1, (0.6123, 0.3711)), (2, (1.913, 0.0))):
        outr = torch.empty_like(x)
        emulate_lincomb(outr, x, y, z if mode == 1 else None, a, b, mode)
        outd = torch.empty_like(x).cuda()
        E.lincomb(outd, x.cuda(), y.cuda(), z.cuda() if mode == 1 else None, a, b, mode)
        assert torch.equal(outd.cpu(), outr), f"mode {mode}"
        xin = x.clone().cuda()                      # in-place form (out aliases x), as the solvers use it
        E.lincomb(xin, xin, y.cuda(), z.cuda() if mode == 1 else None, a, b, mode)
        assert torch.equal(xin.cpu(), outr), f"mode {mode} in place"


def test_ancestral_solvers_run_on_gpu():
    """euler_a / dpm++_2s_a (+cfg++) end to end on the HIP engine: finite, right shapes, and deterministic
    given the device RNG seed."""
    import types
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    for name, lam in (("euler_a", 7.5), ("euler_a_cfg++", 0.6), ("dpm++_2s_a", 7.5), ("dpm++_2s_a_cfg++", 0.6)):
        s = get_solver(name, solver_config=types.SimpleNamespace(num_sampling=6), device="cuda", unet_config=cfg, max_batch=2)
        outs = []
        for _ in range(2):
            torch.cuda.manual_seed(5)
            den, x = s.sample(cfg_guidance=lam, prompt=["bad", ["a cat", "a dog"]], seeds=[1, 2], return_latents=True)
            outs.append((den.float().cpu(), x.float().cpu()))
        assert outs[0][0].shape == (2, 4, 16, 16) and torch.isfinite(outs[0][1]).all()
        assert torch.equal(outs[0][1], outs[1][1]), name
