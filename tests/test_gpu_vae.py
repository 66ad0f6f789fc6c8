"""-m gpu: HIP VAE decoder (csrc/vae.hip) vs the fp32 torch restatement of AutoencoderKL on the CPU,
same seeded weights.  Tolerance: image rel-L2 <= 1e-2 (fp16 storage through ~30 layers + one 512-wide
attention whose scores/probabilities are fp16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw,B", [((16, 16), 2), ((32, 16), 1)])
def test_vae_decode_vs_cpu_reference(hw, B):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd.vae import HipVAE, synth_vae_state_dict
    from oracle.vae_ref import VAERef as TorchVAE
    sd = synth_vae_state_dict(0)
    g = torch.Generator().manual_seed(1)
    z = torch.randn((B, 4) + hw, generator=g) * 0.18215 * 1.5
    hip = HipVAE(0.18215, hw, max_batch=B, state_dict=sd)
    img = hip.decode(z.cuda()).cpu()
    ref = TorchVAE(0.18215, device="cpu", dtype=torch.float32, state_dict=sd).decode(z)
    assert img.shape == ref.shape == (B, 3, 8 * hw[0], 8 * hw[1])
    rel = float((img - ref).norm() / ref.norm())
    assert torch.isfinite(img).all() and rel < 1e-2, f"VAE decode rel-L2 {rel:.3e}"
    # the folded image post-processing of sample(): bit-identical to the reference's two fp32 ops on the decoded image
    assert torch.equal(hip.decode_image(z.cuda()).cpu(), (img / 2 + 0.5).clamp(0, 1))


@pytest.mark.parametrize("hw,B", [((16, 16), 2), ((16, 32), 1)])
def test_vae_encode_vs_cpu_reference(hw, B):
    """HIP encoder (conv_in, asymmetric stride-2 downsamples, mid attention, 8-channel conv_out, quant_conv +
    posterior kernel) vs the fp32 torch restatement; the posterior noise is pinned so the sample is comparable."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from cfgpp_amd.vae import HipVAE, synth_vae_state_dict
    from oracle.vae_ref import VAERef as TorchVAE
    sd = synth_vae_state_dict(0)
    g = torch.Generator().manual_seed(2)
    img = (torch.rand((B, 3, 8 * hw[0], 8 * hw[1]), generator=g) * 2 - 1)
    noise = torch.randn((B, 4) + hw, generator=g)
    hip = HipVAE(0.18215, hw, max_batch=B, state_dict=sd)
    z, mom = hip.encode(img, noise=noise, return_moments=True)
    z_mean = hip.encode(img, sample=False)
    ref = TorchVAE(0.18215, device="cpu", dtype=torch.float32, state_dict=sd)
    mean, logvar = ref.encode_moments(img)
    ref_mom = torch.cat([mean, logvar], dim=1)
    rel_m = float((mom.cpu() - ref_mom).norm() / ref_mom.norm())
    ref_z = (mean + torch.exp(0.5 * logvar) * noise) * 0.18215
    rel_z = float((z.cpu() - ref_z).norm() / ref_z.norm())
    rel_mean = float((z_mean.cpu() - mean * 0.18215).norm() / (mean * 0.18215).norm())
    assert torch.isfinite(z).all() and rel_m < 1e-2 and rel_z < 1e-2 and rel_mean < 1e-2, (rel_m, rel_z, rel_mean)
    # decode(encode(x)) runs through both plans of the same engine (shared activation pool)
    rec = hip.decode(z_mean)
    ref_rec = ref.decode(mean * 0.18215)
    rel_r = float((rec.cpu() - ref_rec).norm() / ref_rec.norm())
    assert rel_r < 2e-2, rel_r


def test_softmax_rows_kernel():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import hip_ops as H
    from cfgpp_amd._lib import check
    for ncols in (256, 4096, 16384):
        x = (torch.randn(37, ncols) * 3).half()
        d = x.clone().cuda()
        check(H.lib().cfgpp_op_softmax_rows(d.data_ptr(), 37, ncols, H.stream()), "softmax")
        ref = torch.softmax(x.float(), dim=-1)
        assert torch.allclose(d.float().cpu(), ref, rtol=2e-3, atol=1e-6), ncols
