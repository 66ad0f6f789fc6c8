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
