"""-m gpu: the engine split into concurrent lanes (cfgpp_amd/hip_engine.py: row groups of the UNet batch on their own HIP
streams) gives the rows the single-stream engine gives, is ordered against the caller's stream by events alone, and passes
the oracle parity of a whole chain."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _engines(cfg, B, hw, lanes_list):
    from cfgpp_amd.hip_engine import HipEngine
    return [HipEngine(cfg, max_batch=B, latent_hw=(hw, hw), lanes=l) for l in lanes_list]


@pytest.mark.parametrize("cfg_name,B,hw,lanes", [("tiny_sd", 4, 16, 2), ("tiny_sd", 3, 16, 4), ("tiny_xl", 2, 16, 2), ("tiny_xl", 2, 16, 4),
                                                 ("tiny_sd", 1, 16, 2)])
def test_lanes_equal_single_stream(cfg_name, B, hw, lanes):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd.unet_config import CONFIGS
    cfg = CONFIGS[cfg_name]
    one, many = _engines(cfg, B, hw, [1, lanes])
    g = torch.Generator().manual_seed(5)
    uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
    c = (torch.randn(B, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
    te = ti = None
    if cfg.addition_embed:
        te = (torch.randn(2 * B, cfg.addition_pooled_dim, generator=g) * 0.5).half().cuda()
        ti = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * (2 * B)).cuda()
    outs = []
    for eng in (one, many):
        eng.set_context(uc, c, te, ti)
        z = torch.randn(B, 4, hw, hw, generator=g.manual_seed(9)).cuda()
        res = []
        for t in (981.0, 401.0, 1.0):
            # the input is produced on the caller's stream right before the call and the output consumed right after it: both
            # orderings are the engine's job (events), there is no synchronize in between
            zz = z * 1.0
            eu, ec = eng.predict(zz, t)
            res.append(torch.cat([eu, ec]).float())
        torch.cuda.synchronize()
        outs.append(torch.stack(res).cpu())
    a, b = outs
    assert bool(torch.isfinite(b).all())
    rel = float((a - b).norm() / a.norm())
    # same kernels on the same rows; only the rule-based K-split launches (tile count follows the row count) may sum in a
    # different order: fp16-rounding-level differences at most
    assert rel < 5e-4, rel


def test_chain_with_lanes_vs_oracle():
    """ddim_cfg++ chain on a 2-lane engine against UNetRef + oracle.sampler (same tolerance as the single-stream chain test)"""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd.hip_engine import HipEngine
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    B, nfe = 2, 20
    sc = types.SimpleNamespace(num_sampling=nfe)
    eng = HipEngine(cfg, max_batch=B, lanes=2)
    hip = get_solver("ddim_cfg++", solver_config=sc, device="cuda", unet_config=cfg, max_batch=B, engine=eng, scalar_semantics="cuda")
    uc, c = hip.get_text_embed("bad", ["a cat", "a dog"])
    net = UNetRef(cfg, synth_state_dict(cfg, 0))
    ref = get_solver("ddim_cfg++", solver_config=sc, device="cpu", unet_config=cfg, max_batch=B, text_encoder=hip.text_encoder,
                     engine=MockEngine(lambda z, t, ehs, te, ti: net(z, t, ehs.float())["sample"].half(), (16, 16)), scalar_semantics="cuda")
    kw = dict(cfg_guidance=0.6, prompt_embeds=(uc, c), seeds=[11, 12], return_latents=True)
    a = hip.sample(**kw)[0].float().cpu()
    kw["prompt_embeds"] = (uc.cpu(), c.cpu())
    b = ref.sample(**kw)[0].float()
    rel = float((a - b).norm() / b.norm())
    assert rel < 3e-3, rel


def test_check_finite_guard_names_the_timestep(monkeypatch):
    """CFGPP_CHECK_FINITE=1: a forward whose output is not finite stops the job with the timestep (here: an fp32 latent far
    outside the fp16 range of conv_in's output)"""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd._lib import CfgppError
    from cfgpp_amd.hip_engine import HipEngine
    from cfgpp_amd.unet_config import TINY_SD as cfg
    monkeypatch.setenv("CFGPP_CHECK_FINITE", "1")
    eng = HipEngine(cfg, max_batch=1)
    assert eng.check_finite
    g = torch.Generator().manual_seed(0)
    uc = (torch.randn(1, 77, cfg.cross_attention_dim, generator=g) * 0.5).half().cuda()
    eng.set_context(uc, uc)
    z = torch.randn(1, 4, 16, 16, generator=g).cuda()
    eng.predict(z, 500.0)                         # finite input: passes
    with pytest.raises(CfgppError, match="t=321"):
        eng.predict(z * 3e38, 321.0)
