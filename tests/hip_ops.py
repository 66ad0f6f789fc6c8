"""Thin test-side wrappers that call the single-op C-ABI entry points of
libcfgpp_hip.so on torch CUDA tensors, plus layout helpers (halo-padded NHWC,
head-major Q/K/V^T) and the repacking the engine does at finalize()."""
from __future__ import annotations

import torch

from cfgpp_amd import _lib
from cfgpp_amd._lib import check

DEV = "cuda"


def lib():
    return _lib.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


def P(t):
    return None if t is None else t.data_ptr()


def round_up(a, b):
    return (a + b - 1) // b * b


# ---- layouts ----------------------------------------------------------------
def to_pn(x_nchw: torch.Tensor) -> torch.Tensor:
    """NCHW (any float) -> halo-padded NHWC fp16 on the GPU."""
    N, C, H, W = x_nchw.shape
    out = torch.zeros((N, H + 2, W + 2, C), dtype=torch.float16, device=DEV)
    out[:, 1:H + 1, 1:W + 1, :] = x_nchw.permute(0, 2, 3, 1).to(DEV, torch.float16)
    return out


def empty_pn(N, H, W, C) -> torch.Tensor:
    return torch.zeros((N, H + 2, W + 2, C), dtype=torch.float16, device=DEV)


def from_pn(pn: torch.Tensor) -> torch.Tensor:
    """halo-padded NHWC -> NCHW fp32 CPU (interior only)."""
    N, Hp, Wp, C = pn.shape
    return pn[:, 1:Hp - 1, 1:Wp - 1, :].permute(0, 3, 1, 2).float().cpu()


def halo_is_zero(pn: torch.Tensor) -> bool:
    a = pn.float()
    return bool((a[:, 0].abs().sum() + a[:, -1].abs().sum() + a[:, :, 0].abs().sum() + a[:, :, -1].abs().sum()) == 0)


def pack_conv3(w_oihw: torch.Tensor) -> torch.Tensor:
    """OIHW -> [O][I/64][9][64] fp16 (k = (cb*9 + tap)*64 + c: channel-block major, tap minor)."""
    O, I, kh, kw = w_oihw.shape
    return (w_oihw.permute(0, 2, 3, 1).reshape(O, kh * kw, I // 64, 64).permute(0, 2, 1, 3)
            .reshape(O, kh * kw * I).to(DEV, torch.float16).contiguous())


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """[8C, C] -> packed rows: per 64 packed rows, [0,32) value features, [32,64) gate features."""
    F4 = w.shape[0] // 2
    f = torch.arange(F4)
    pv = (f // 32) * 64 + (f % 32)
    wp = torch.empty_like(w)
    bp = torch.empty_like(b)
    wp[pv] = w[:F4]
    wp[pv + 32] = w[F4:]
    bp[pv] = b[:F4]
    bp[pv + 32] = b[F4:]
    return wp.to(DEV, torch.float16).contiguous(), bp.to(DEV, torch.float32).contiguous()


# ---- ops ----------------------------------------------------------------------
def igemm(a0, a1, C0, C1, taps, amode, H, W, w, M, N, bias=None, temb=None, temb_ld=0, resid=None, rmode=0, rld=0,
          out=None, omode=0, old=0, epi=0):
    check(lib().cfgpp_op_igemm(P(a0), P(a1), C0, C1, taps, amode, H, W, P(w), M, N, P(bias), P(temb), temb_ld,
                               P(resid), rmode, rld, P(out), omode, old, epi, stream()), "cfgpp_op_igemm")
    return out


def linear(a, w, bias=None, resid=None, epi=0):
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N // 2 if epi == 1 else N), dtype=torch.float16, device=DEV)
    igemm(a, None, K, 0, 1, 0, 0, 0, w, M, N, bias=bias, resid=resid, rmode=0, rld=N, out=out, omode=0,
          old=out.shape[1], epi=epi)
    return out


def conv3x3(x_pn, w_packed, bias, Hout, Wout, amode=1, temb=None, temb_ld=0, resid_pn=None):
    N = x_pn.shape[0]
    Cin = x_pn.shape[3]
    Cout = w_packed.shape[0]
    out = empty_pn(N, Hout, Wout, Cout)
    igemm(x_pn, None, Cin, 0, 9, amode, Hout, Wout, w_packed, N * Hout * Wout, Cout, bias=bias, temb=temb,
          temb_ld=temb_ld, resid=resid_pn, rmode=1, rld=Cout, out=out, omode=1, old=Cout, epi=0)
    return out


def conv1x1_2src(x0_pn, x1_pn, w, bias):
    N, Hp, Wp, C0 = x0_pn.shape
    H, W = Hp - 2, Wp - 2
    C1 = 0 if x1_pn is None else x1_pn.shape[3]
    Cout = w.shape[0]
    out = empty_pn(N, H, W, Cout)
    igemm(x0_pn, x1_pn, C0, C1, 1, 1, H, W, w, N * H * W, Cout, bias=bias, out=out, omode=1, old=Cout)
    return out


def groupnorm(x0_pn, x1_pn, gamma, beta, G, eps, silu, dst_padded=True):
    N, Hp, Wp, C0 = x0_pn.shape
    H, W = Hp - 2, Wp - 2
    C1 = 0 if x1_pn is None else x1_pn.shape[3]
    C = C0 + C1
    stats = torch.zeros(N * (1024 * G * 2 + G * 2), dtype=torch.float32, device=DEV)
    if dst_padded:
        dst = empty_pn(N, H, W, C)
    else:
        dst = torch.empty((N * H * W, C), dtype=torch.float16, device=DEV)
    check(lib().cfgpp_op_groupnorm(P(x0_pn), P(x1_pn), P(dst), P(gamma), P(beta), P(stats), N, H, W, C0, C1, G,
                                   float(eps), int(silu), int(dst_padded), stream()), "cfgpp_op_groupnorm")
    return dst


def groupnorm_pre(x0_pn, x1_pn, gst0, gst1, gamma, beta, G, eps, silu, dst_padded=True):
    """GroupNorm with the statistics its inputs' producers left behind (cfgpp_op_igemm_set_gstat)"""
    N, Hp, Wp, C0 = x0_pn.shape
    H, W = Hp - 2, Wp - 2
    C1 = 0 if x1_pn is None else x1_pn.shape[3]
    stats = torch.zeros(N * G * 2, dtype=torch.float32, device=DEV)
    dst = empty_pn(N, H, W, C0 + C1) if dst_padded else torch.empty((N * H * W, C0 + C1), dtype=torch.float16, device=DEV)
    check(lib().cfgpp_op_groupnorm_pre(P(x0_pn), P(x1_pn), P(dst), P(gamma), P(beta), P(gst0), P(gst1), P(stats), N, H, W, C0, C1, G,
                                       float(eps), int(silu), int(dst_padded), stream()), "cfgpp_op_groupnorm_pre")
    return dst


def layernorm(x, gamma, beta, eps=1e-5):
    y = torch.empty_like(x)
    check(lib().cfgpp_op_layernorm(P(x), P(y), P(gamma), P(beta), x.shape[0], x.shape[1], float(eps), stream()),
          "cfgpp_op_layernorm")
    return y


def heads_project(a, w, B, tokens, C, nheads, part0, nparts, q_pad, k_pad):
    """a [B*tokens, K] x w [nparts*C, K] -> head-major buffers (zero initialised)."""
    d = C // nheads
    dp = round_up(d, 32)
    hq = torch.zeros((B * nheads, q_pad, dp), dtype=torch.float16, device=DEV)
    hk = torch.zeros((B * nheads, k_pad, dp), dtype=torch.float16, device=DEV)
    hvt = torch.zeros((B * nheads, dp, k_pad), dtype=torch.float16, device=DEV)
    check(lib().cfgpp_op_attention_prepare_vt(P(hvt), B * nheads, d, k_pad, stream()), "cfgpp_op_attention_prepare_vt")
    check(lib().cfgpp_op_igemm_heads(P(a), a.shape[1], P(w), a.shape[0], w.shape[0], None, tokens, P(hq), P(hk),
                                     P(hvt), part0, C, d, nheads, q_pad, k_pad, stream()), "cfgpp_op_igemm_heads")
    return hq, hk, hvt


def vt_pos(n):
    """column of key 0..n-1 in a V^T buffer (bits 2 and 3 of the key index swapped; include/cfgpp.h)"""
    key = torch.arange(n)
    return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1)


def make_heads(q, k, v):
    """q [B,h,Nq,d], k/v [B,h,Nk,d] (cpu float) -> padded head-major device buffers."""
    B, h, Nq, d = q.shape
    Nk = k.shape[2]
    dp = round_up(d, 32)
    q_pad, k_pad = round_up(Nq, 128), round_up(Nk, 64)
    hq = torch.zeros((B * h, q_pad, dp), dtype=torch.float16, device=DEV)
    hk = torch.zeros((B * h, k_pad, dp), dtype=torch.float16, device=DEV)
    hvt = torch.zeros((B * h, dp, k_pad), dtype=torch.float16, device=DEV)
    hq[:, :Nq, :d] = q.reshape(B * h, Nq, d).to(DEV, torch.float16)
    hk[:, :Nk, :d] = k.reshape(B * h, Nk, d).to(DEV, torch.float16)
    # V^T contract: within every 32-key block the key index has bits 2 and 3 swapped (include/cfgpp.h)
    hvt[:, :d, vt_pos(Nk).to(DEV)] = v.reshape(B * h, Nk, d).transpose(1, 2).to(DEV, torch.float16)
    check(lib().cfgpp_op_attention_prepare_vt(P(hvt), B * h, d, k_pad, stream()), "cfgpp_op_attention_prepare_vt")   # ones row when d % 32 != 0
    return hq, hk, hvt, q_pad, k_pad


def attention(hq, hk, hvt, B, nheads, d, nq, nk, q_pad, k_pad):
    o = torch.empty((B, nq, nheads * d), dtype=torch.float16, device=DEV)
    check(lib().cfgpp_op_attention(P(hq), P(hk), P(hvt), P(o), B, nheads, d, nq, nk, q_pad, k_pad, stream()),
          "cfgpp_op_attention")
    return o


def conv_in(z, w_oihw, bias, R):
    zB, Cin, H, W = z.shape
    Cout = w_oihw.shape[0]
    wk = w_oihw.permute(2, 3, 1, 0).reshape(9 * Cin, Cout).to(DEV, torch.float32).contiguous()
    out = empty_pn(R, H, W, Cout)
    check(lib().cfgpp_op_conv_in(P(z), int(z.dtype == torch.float16), P(out), P(wk), P(bias), R, zB, Cin, H, W, Cout,
                                 stream()), "cfgpp_op_conv_in")
    return out


def conv_out(x_pn, w_oihw, bias, out_half=True):
    R, Hp, Wp, C = x_pn.shape
    H, W = Hp - 2, Wp - 2
    Co = w_oihw.shape[0]
    wk = w_oihw.permute(0, 2, 3, 1).reshape(Co, 9, C).to(DEV, torch.float16).contiguous()
    out = torch.empty((R, Co, H, W), dtype=torch.float16 if out_half else torch.float32, device=DEV)
    check(lib().cfgpp_op_conv_out(P(x_pn), P(out), int(out_half), P(wk), P(bias), R, H, W, C, Co, stream()),
          "cfgpp_op_conv_out")
    return out


def sinusoid(vals, dim):
    out = torch.empty((vals.numel(), dim), dtype=torch.float32, device=DEV)
    check(lib().cfgpp_op_sinusoid(P(vals), 0.0, P(out), vals.numel(), dim, dim, 0, stream()), "cfgpp_op_sinusoid")
    return out


def skinny(x, w, bias, silu_in=False, silu_out=False, addend=None):
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    check(lib().cfgpp_op_skinny_gemm(P(x), K, P(w), P(bias), P(addend), N if addend is not None else 0, P(out), N, M, N,
                                     K, int(silu_in), int(silu_out), stream()), "cfgpp_op_skinny_gemm")
    return out


def err_stats(got: torch.Tensor, ref: torch.Tensor) -> dict:
    g = got.detach().float().cpu()
    r = ref.detach().float().cpu()
    diff = (g - r)
    return dict(rel_l2=float(diff.norm() / (r.norm() + 1e-30)), max_abs=float(diff.abs().max()),
                ref_max=float(r.abs().max()), finite=bool(torch.isfinite(g).all()))
