"""The CPU oracle (oracle/sampler.py) must reproduce, bit for bit, what the
reference's own sampler code produced (tests/golden/sampler_golden.npz, recorded by
tests/golden/make_golden.py from /root/reference behind a stub `diffusers`)."""
import numpy as np
import pytest
import torch

from cfgpp_amd.schedule import SchedulerTables
from oracle import sampler as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


DDIM_CASES = [
    # tag, nfe, kind, lam, cfgpp, wrap
    ("G2/sd_ddim_cfgpp_h", 50, "ddim", 0.6, True, False),
    ("G2/sd_ddim_cfgpp_f", 50, "ddim", 0.6, True, False),
    ("G2/sd_ddim_cfg_h", 10, "ddim", 7.5, False, False),
    ("G3/sd_inv_cfgpp", 10, "ddim", 0.6, True, False),
    ("G3/sd_inv_cfg", 10, "ddim", 2.0, False, False),
    ("G3/sd_edit_cfgpp", 10, "ddim", 0.6, True, False),
    ("G2/xl_ddim_cfgpp", 50, "ddim", 0.6, True, True),
    ("G2/xl_ddim_cfg", 10, "ddim", 5.0, False, True),
    ("G2/xl_light_ddim_cfgpp", 4, "lightning", 1.0, True, True),
    ("G3/xl_edit_cfgpp", 10, "ddim", 0.6, True, False),
    ("G3/xl_edit_cfgpp_recon", 10, "ddim", 0.6, True, False),
    ("G3/xl_edit_cfg", 10, "ddim", 3.0, False, False),
]


@pytest.mark.parametrize("tag,nfe,kind,lam,cfgpp,wrap", DDIM_CASES)
def test_ddim_steps_bit_exact(golden, tag, nfe, kind, lam, cfgpp, wrap):
    g, _ = golden
    tb = SchedulerTables(nfe, kind)
    z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
    n = z0.shape[0]
    off = z.shape[0] - n          # leading inversion calls
    ts = tb.timesteps.int() if wrap else tb.timesteps
    assert n == len(ts)
    for i, t in enumerate(ts):
        a, b = O.ddim_step(z[off + i][0:1], e[off + i][0:1], e[off + i][1:2], lam, None, None, False, cfgpp,
                           sqrt4=tb.ddim_sqrt_coeffs(t, wrap=wrap))
        assert torch.equal(a, z0[i]) and torch.equal(b, zt[i]), f"{tag} step {i}"
    for i, t in enumerate(reversed(tb.timesteps)):
        if i >= off:
            break
        _, b = O.ddim_step(z[i][0:1], e[i][0:1], e[i][1:2], lam, None, None, cfgpp, False,
                           sqrt4=tb.ddim_sqrt_coeffs(t, inversion=True))
        assert torch.equal(b, z[i + 1][0:1]), f"{tag} inversion step {i}"


def test_pinned_sqrt_tables_match_torch_here():
    """The pinned sqrt tables equal torch's own evaluation in the recording container;
    on another host torch.sqrt may differ by 1 ulp (that is why they are pinned)."""
    tb = SchedulerTables(50)
    a = tb.alphas_cumprod
    d0 = int((a.sqrt() != tb._sqrt_a).sum())
    d1 = int(((1 - a).sqrt() != tb._sqrt_1ma).sum())
    assert d0 <= 16 and d1 <= 16            # identical here (0); tolerate a few ulp-flips elsewhere
    assert torch.allclose(a.sqrt(), tb._sqrt_a, rtol=2e-7, atol=0) and torch.allclose((1 - a).sqrt(), tb._sqrt_1ma, rtol=2e-7, atol=1e-9)


KDIFF_CASES = [
    ("G4/sd_dpm2m_cfgpp", 20, 0.6, "cfgpp_sd", "sd"),
    ("G4/sd_dpm2m_cfg", 10, 7.5, "cfg", "sd"),
    ("G4/sd_euler_cfgpp", 10, 0.6, "euler_cfgpp", "sd"),
    ("G4/sd_euler_cfg", 10, 7.5, "euler_cfg", "sd"),
    ("G4/xl_dpm2m_cfgpp", 20, 0.6, "cfgpp_xl", "xl"),
    ("G4/xl_light_dpm2m_cfgpp", 4, 1.0, "cfgpp_xl", "xl"),
]


@pytest.mark.parametrize("tag,nfe,lam,variant,kind", KDIFF_CASES)
def test_kdiff_steps_bit_exact(golden, tag, nfe, lam, variant, kind):
    g, _ = golden
    tb = SchedulerTables(nfe, "lightning" if "light" in tag else "ddim")
    z, e, z0, zt, ut = (T(g[tag + k]) for k in ("/unet_z", "/unet_eps", "/z0t", "/zt", "/unet_t"))
    if kind == "sd":
        sig = tb.karras_sigmas()
        torch.manual_seed(42)
        x = (torch.randn(1, 4, 64, 64) * (sig[0] ** 2 + 1) ** 0.5).to(torch.float16)[..., :8, :8]
        n = nfe
    else:
        alphas = tb.alphas_cumprod[tb.timesteps.int()]
        sig = (1 - alphas).sqrt() / alphas.sqrt()
        torch.manual_seed(42)
        x = torch.randn(1, 4, 8, 8).to(torch.float16) * sig[0]
        n = nfe - 1
    assert z0.shape[0] == n
    old = None
    for i in range(n):
        if kind == "sd":
            xc, tq = O.kdiff_input_div(x, sig[i]), int(tb.timestep(sig[i]))
        else:
            xc, tq = O.kdiff_input_mul(x, alphas[i].clone().sqrt()), int(tb.sigma_to_t(sig[i]))
        assert torch.equal(xc, z[i][0:1]) and tq == int(ut[i][0])
        den, uden = O.kdiff_denoised(x, e[i][0:1], e[i][1:2], lam, sig[i], xl_form=(kind == "xl"))
        if variant.startswith("euler"):
            xn = O.euler_step(x, den, uden if variant == "euler_cfgpp" else den, sig[i], sig[i + 1])
        else:
            xn, old = O.dpm2m_step(x, den, uden, old, sig, i, variant)
        assert torch.equal(den, z0[i]) and torch.equal(xn, zt[i]), f"{tag} step {i}"
        x = zt[i]


def test_whole_loop_driver_matches_golden(golden):
    """oracle.sample_ddim / invert_ddim (used by bench.py's cpu_baseline leg) end to end."""
    from _stub_env import fake_embed, pointwise_eps
    g, meta = golden
    tag = "G2/sd_ddim_cfgpp_h"
    null, prompt = meta[tag]["prompts"]
    uc, c = fake_embed("L" + null, (1, 77, 768)), fake_embed("L" + prompt, (1, 77, 768))
    ehs = torch.cat([uc, c])
    torch.manual_seed(42)
    zT = torch.randn(1, 4, 64, 64)[..., :8, :8].contiguous()
    tb = SchedulerTables(50)

    def unet(z, t):
        eps = pointwise_eps(torch.cat([z, z]), torch.as_tensor(float(t)).reshape(1), ehs)
        return eps[:1], eps[1:]
    z0t, zt = O.sample_ddim(unet, zT, tb, 0.6, cfgpp=True)
    assert torch.allclose(z0t, T(g[tag + "/z0t"])[-1], rtol=0, atol=2e-3)
    assert torch.allclose(zt, T(g[tag + "/zt"])[-1], rtol=0, atol=2e-3)
