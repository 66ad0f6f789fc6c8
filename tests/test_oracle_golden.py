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


H16_CASES = [("G3h/sd_inv_cfgpp", 0.6, True), ("G3h/sd_inv_cfg", 2.0, False), ("G3h/sd_edit_cfgpp", 0.6, True),
             ("G3h/xl_edit_cfgpp", 0.6, True), ("G3h/xl_edit_cfgpp_recon", 0.6, True), ("G3h/xl_edit_cfg", 3.0, False)]


@pytest.mark.parametrize("tag,lam,cfgpp", H16_CASES)
def test_ddim_steps_fp16_latent_bit_exact(golden_h16, tag, lam, cfgpp):
    """fp16 VAE latent -> the whole inversion + regeneration chain is fp16 (every op rounds to fp16)"""
    g, meta = golden_h16
    assert meta[tag + "/dtypes"] == ["torch.float16", "torch.float16"]
    tb = SchedulerTables(10)
    z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
    assert z.dtype == torch.float16 and z0.dtype == torch.float16
    n = z0.shape[0]
    off = z.shape[0] - n
    for i, t in enumerate(reversed(tb.timesteps)):          # inversion
        _, b = O.ddim_step(z[i][0:1], e[i][0:1], e[i][1:2], lam, None, None, cfgpp, False,
                           sqrt4=tb.ddim_sqrt_coeffs(t, inversion=True))
        assert b.dtype == torch.float16 and torch.equal(b, z[i + 1][0:1]), f"{tag} inversion step {i}"
    for i, t in enumerate(tb.timesteps):                    # regeneration
        a, b = O.ddim_step(z[off + i][0:1], e[off + i][0:1], e[off + i][1:2], lam, None, None, False, cfgpp,
                           sqrt4=tb.ddim_sqrt_coeffs(t))
        assert torch.equal(a, z0[i]) and torch.equal(b, zt[i]), f"{tag} step {i}"


def test_cuda_scalar_semantics_differs_only_by_scalar_rounding():
    """"cuda" semantics (the product default) = the same formulas with the scalar-first coefficients left in fp32 and
    the scalar divisor applied as a host-side fp32 reciprocal.  Here (no GPU) only that the oracle says what it claims;
    the pin against torch-ROCm evaluating the reference's expressions is tests/test_gpu_torch_semantics.py."""
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 4, 8, 8, generator=g)
    eu, ec = (torch.randn(1, 4, 8, 8, generator=g).half() for _ in range(2))
    tb = SchedulerTables(50)
    s4 = tb.ddim_sqrt_coeffs(tb.timesteps[3])
    a0, b0 = O.ddim_step(z, eu, ec, 0.6, None, None, False, True, sqrt4=s4, semantics="cpu")
    a1, b1 = O.ddim_step(z, eu, ec, 0.6, None, None, False, True, sqrt4=s4, semantics="cuda")
    hat = O.cfg_mix(eu, ec, 0.6)
    c1, c2, c3, c4 = (torch.tensor(float(v)) for v in s4)
    want_a = (z - (hat.float() * c1).half().float()) * (torch.tensor(1.0) / c2)
    want_b = c3 * want_a + (eu.float() * c4).half().float()
    assert torch.equal(a1, want_a) and torch.equal(b1, want_b)
    assert not torch.equal(a0, a1) and float((a0 - a1).abs().max()) < 2e-3 * float(a1.abs().max())


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


def test_realsize_fixtures_are_present_and_shaped():
    """the real-size oracle outputs recorded by tests/golden/make_unet_golden.py (used by the -m gpu real-size tests)"""
    import json
    import os

    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(here, "realsize_golden.json")))
    want = {"sd15_fwd": {"eps": ([16, 4, 64, 64], "float16")}, "sdxl_fwd": {"eps": ([4, 4, 128, 128], "float16")},
            "sd15_chain": {"z0t": ([8, 4, 64, 64], "float32")},
            "sdxl_chain": {"ddim_cfg++": ([2, 4, 128, 128], "float32"), "ddim_cfg++_lightning": ([1, 4, 128, 128], "float32")},
            "sd15_fwd_r2": {"t981": ([2, 4, 64, 64], "float16"), "t1": ([2, 4, 64, 64], "float16")},
            "sdxl_fwd_r2_32": {"t981": ([2, 4, 32, 32], "float16"), "t1": ([2, 4, 32, 32], "float16")}}
    for case, arrays in want.items():
        with np.load(os.path.join(here, f"realsize_{case}.npz")) as f:
            assert set(f.files) == set(arrays), case
            for k, (shape, dt) in arrays.items():
                a = f[k]
                assert list(a.shape) == shape and str(a.dtype) == dt and np.isfinite(a.astype(np.float32)).all(), (case, k)
                assert 0.05 < float(np.abs(a.astype(np.float32)).mean()) < 100.0, (case, k)     # a real eps / latent, not zeros (the 1-NFE Lightning z0t divides by sqrt(alpha_999) = 0.068)
                assert meta[case]["arrays"][k] == [shape, dt]
