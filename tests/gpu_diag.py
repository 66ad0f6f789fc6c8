#!/usr/bin/env python
"""One-shot GPU diagnostic: every kernel against its fp32 torch reference, error
statistics + micro-benchmarks written to gpurun_out/diag.json.  Never raises:
each case is isolated so one broken kernel does not hide the others.

    python tests/gpu_diag.py [--quick] [--full-unet]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import traceback

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hip_ops as H  # noqa: E402
from cfgpp_amd import _lib  # noqa: E402

RESULTS = {}


def case(name):
    def deco(fn):
        def run(*a, **k):
            t0 = time.time()
            try:
                r = fn(*a, **k)
                torch.cuda.synchronize()
                RESULTS[name] = r
                print(f"[diag] {name}: {json.dumps(r)}  ({time.time() - t0:.1f}s)", flush=True)
            except Exception as e:  # noqa: BLE001
                RESULTS[name] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
                print(f"[diag] {name}: ERROR {type(e).__name__}: {e}", flush=True)
        return run
    return deco


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().float()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


# ----------------------------------------------------------------------------
@case("gemm_basic")
def t_gemm(cfgs=(0, 1, 2, 3), M=300, N=192, K=128):
    out = {}
    a = rnd(M, K, seed=1)
    w = rnd(N, K, scale=K ** -0.5, seed=2)
    b = rnd(N, scale=0.1, seed=3)
    r = rnd(M, N, seed=4)
    ref = a @ w.t() + b + r
    for c in cfgs:
        H.lib().cfgpp_igemm_force_config(c)
        got = H.linear(a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16), b.to(H.DEV), r.to(H.DEV, torch.float16))
        out[f"cfg{c}"] = H.err_stats(got, ref)
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("gemm_transpose_check")
def t_gemm_t():
    # A = I-like asymmetric check: out[m][n] = W[n][m] when A = identity (K = M)
    M = K = 128
    N = 64
    a = torch.eye(M)
    w = (torch.arange(N * K).reshape(N, K) % 251).float() / 64.0
    got = H.linear(a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16))
    return H.err_stats(got, w.t().half().float())


@case("gemm_geglu")
def t_geglu(M=260, C=64):
    a = rnd(M, C, seed=5)
    w = rnd(8 * C, C, scale=C ** -0.5, seed=6)
    b = rnd(8 * C, scale=0.1, seed=7)
    h = a @ w.t() + b
    v, g = h.chunk(2, dim=-1)
    ref = v * F.gelu(g)
    wp, bp = H.pack_geglu(w, b)
    out = {}
    for c in (1, 2, 4, 6, 10, 12, 13, 14, 15, 16, 17, 20, 24, 26):
        H.lib().cfgpp_igemm_force_config(c)
        got = H.linear(a.to(H.DEV, torch.float16), wp, bp, epi=1)
        out[f"cfg{c}"] = H.err_stats(got, ref)
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("conv3x3")
def t_conv(N=2, Cin=64, Cout=128, Hh=12, Ww=10):
    x = rnd(N, Cin, Hh, Ww, seed=8)
    w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=9)
    b = rnd(Cout, scale=0.1, seed=10)
    temb = rnd(N, Cout, scale=0.5, seed=11)
    res = rnd(N, Cout, Hh, Ww, seed=12)
    ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + res
    out = {}
    for c in (1, 2, 3):
        H.lib().cfgpp_igemm_force_config(c)
        got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), Hh, Ww, 1, temb.to(H.DEV), Cout, H.to_pn(res))
        out[f"cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got))
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("conv3x3_temb_small_maps")
def t_conv_temb_small():
    """time-embedding epilogue on feature maps under 8 x 8 (a tile's rows span many batches: the launcher sizes the LDS
    parameter segments for every batch a tile touches, or drops to the 64 x 64 tile when they do not fit), for every tile
    family"""
    out = {}
    for (N, Hh, Ww, Cout) in ((40, 4, 4, 320), (70, 2, 2, 320), (9, 2, 3, 128), (3, 8, 8, 320)):
        Cin = 64
        x = rnd(N, Cin, Hh, Ww, seed=300 + N)
        w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=301 + N)
        b = rnd(Cout, scale=0.1, seed=302 + N)
        temb = rnd(N, Cout, scale=0.5, seed=303 + N)
        res = rnd(N, Cout, Hh, Ww, seed=304 + N)
        ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + res
        for c in (0, 1, 3, 4, 5, 6, 7, 11, 13, 14, 15, 16, 17, 19, 20):
            H.lib().cfgpp_igemm_force_config(c)
            got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), Hh, Ww, 1, temb.to(H.DEV), Cout, H.to_pn(res))
            out[f"n{N}_{Hh}x{Ww}_cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got))
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("conv3x3_stride2")
def t_conv_s2(N=2, C=64, Hh=12, Ww=8):
    x = rnd(N, C, Hh, Ww, seed=13)
    w = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=14)
    b = rnd(C, scale=0.1, seed=15)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), Hh // 2, Ww // 2, 2)
    return H.err_stats(H.from_pn(got), ref)


@case("conv3x3_upsample")
def t_conv_up(N=2, C=64, Hh=6, Ww=5):
    x = rnd(N, C, Hh, Ww, seed=16)
    w = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=17)
    b = rnd(C, scale=0.1, seed=18)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), 2 * Hh, 2 * Ww, 3)
    return H.err_stats(H.from_pn(got), ref)


@case("conv1x1_two_sources")
def t_conv1(N=2, C0=128, C1=64, Cout=64, Hh=9, Ww=7):
    x0, x1 = rnd(N, C0, Hh, Ww, seed=19), rnd(N, C1, Hh, Ww, seed=20)
    w = rnd(Cout, C0 + C1, scale=(C0 + C1) ** -0.5, seed=21)
    b = rnd(Cout, scale=0.1, seed=22)
    ref = F.conv2d(torch.cat([x0, x1], 1), w[:, :, None, None], b)
    got = H.conv1x1_2src(H.to_pn(x0), H.to_pn(x1), w.to(H.DEV, torch.float16), b.to(H.DEV))
    return H.err_stats(H.from_pn(got), ref)


@case("igemm_big_tiles")
def t_big():
    """8-wave tiles (256x256, 256x320, 256x128): conv3x3 + bias + temb + residual, and a plain linear, vs fp32 torch"""
    out = {}
    x = rnd(2, 128, 24, 20, seed=80)
    w = rnd(320, 128, 3, 3, scale=(9 * 128) ** -0.5, seed=81)
    b = rnd(320, scale=0.1, seed=82)
    temb = rnd(2, 320, scale=0.5, seed=83)
    res = rnd(2, 320, 24, 20, seed=84)
    ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + res
    a = rnd(700, 256, seed=85)
    wl = rnd(640, 256, scale=1 / 16, seed=86)
    bl = rnd(640, scale=0.1, seed=87)
    refl = a @ wl.t() + bl
    # short K (1 .. 3 K-tiles): fewer tiles than the deeper rings have stages
    shortk = []
    for K in (64, 128, 192):
        ak, wk = rnd(300, K, seed=88 + K), rnd(320, K, scale=K ** -0.5, seed=89 + K)
        shortk.append((K, ak, wk, ak @ wk.t()))
    # 7 / 8: 128x160 / 128x320; 10: 256x320 with the waves stacked along M; 9 / 11 / 12 / 14: 3- and 4-stage LDS rings;
    # 18 / 19: 128x160 as 8 waves on the 16x16x32 MFMA (3 / 4 stages); 15 / 16 / 17: lin32_kernel (32-deep K-tiles, several
    # workgroups per CU) and 13 / 20 (256 x 256 / 256 x 320 on a four-stage ring): tile32_kernel, 32-deep K-tiles
    for c in (4, 5, 6, 7, 8, 10, 9, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 24, 25, 26, 27):
        H.lib().cfgpp_igemm_force_config(c)
        got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), 24, 20, 1, temb.to(H.DEV), 320, H.to_pn(res))
        out[f"conv_cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got))
        out[f"linear_cfg{c}"] = H.err_stats(H.linear(a.to(H.DEV, torch.float16), wl.to(H.DEV, torch.float16), bl.to(H.DEV)), refl)
        if c in (9, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 24, 25, 26, 27):
            for K, ak, wk, rk in shortk:
                out[f"linear_k{K}_cfg{c}"] = H.err_stats(H.linear(ak.to(H.DEV, torch.float16), wk.to(H.DEV, torch.float16)), rk)
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("igemm_tail_split")
def t_tail():
    """K-split tail path (few tiles, long K): conv3x3 + residual, GEGLU and heads epilogues through
    igemm_reduce_kernel; must agree with the unsplit path to fp32-summation-order noise."""
    out = {}
    x = rnd(2, 256, 20, 16, seed=60)
    w = rnd(192, 256, 3, 3, scale=(9 * 256) ** -0.5, seed=61)
    b = rnd(192, scale=0.1, seed=62)
    res = rnd(2, 192, 20, 16, seed=63)
    ref = F.conv2d(x, w, b, padding=1) + res
    for on in (1, 0):
        H.lib().cfgpp_igemm_set_tail_split(on)
        for c in (1, 2):
            H.lib().cfgpp_igemm_force_config(c)
            got = H.conv3x3(H.to_pn(x), H.pack_conv3(w), b.to(H.DEV), 20, 16, 1, None, 0, H.to_pn(res))
            out[f"conv_tail{on}_cfg{c}"] = H.err_stats(H.from_pn(got), ref)
    H.lib().cfgpp_igemm_set_tail_split(1)
    # forced K-split of the 8-wave / 128x160 tiles (the big-tile rule's path): conv + bias + residual, N = 320
    w3 = rnd(320, 256, 3, 3, scale=(9 * 256) ** -0.5, seed=69)
    b3 = rnd(320, scale=0.1, seed=70)
    res3 = rnd(2, 320, 20, 16, seed=71)
    ref3 = F.conv2d(x, w3, b3, padding=1) + res3
    for c, S in ((8, 2), (5, 4), (7, 3), (4, 2), (14, 3), (12, 2)):      # 14: the tile the K-split rule uses
        H.lib().cfgpp_igemm_force_config(c); H.lib().cfgpp_igemm_force_split(S)
        got = H.conv3x3(H.to_pn(x), H.pack_conv3(w3), b3.to(H.DEV), 20, 16, 1, None, 0, H.to_pn(res3))
        out[f"conv_cfg{c}_split{S}"] = dict(H.err_stats(H.from_pn(got), ref3), halo_zero=H.halo_is_zero(got))
    H.lib().cfgpp_igemm_force_split(0)
    a = rnd(300, 2048, seed=64)
    wg = rnd(8 * 64, 2048, scale=2048 ** -0.5, seed=65)
    bg = rnd(8 * 64, scale=0.1, seed=66)
    h = a @ wg.t() + bg
    v, g = h.chunk(2, dim=-1)
    wp, bp = H.pack_geglu(wg, bg)
    for c in (1, 2):
        H.lib().cfgpp_igemm_force_config(c)
        out[f"geglu_cfg{c}"] = H.err_stats(H.linear(a.to(H.DEV, torch.float16), wp, bp, epi=1), v * F.gelu(g))
    H.lib().cfgpp_igemm_force_config(1)
    B, tokens, C, nheads = 2, 160, 128, 2
    a2 = rnd(B * tokens, 2048, seed=67)
    w2 = rnd(3 * C, 2048, scale=2048 ** -0.5, seed=68)
    qp, kp = H.round_up(tokens, 128), H.round_up(tokens, 64)
    hq, hk, hvt = H.heads_project(a2.to(H.DEV, torch.float16), w2.to(H.DEV, torch.float16), B, tokens, C, nheads, 0, 3, qp, kp)
    y = (a2 @ w2.t()).reshape(B, tokens, 3, nheads, C // nheads)
    out["heads_q"] = H.err_stats(hq[:, :tokens, :C // nheads].reshape(B, nheads, tokens, -1), y[:, :, 0].permute(0, 2, 1, 3))
    out["heads_vt"] = H.err_stats(hvt[:, :C // nheads, H.vt_pos(tokens).to(H.DEV)].reshape(B, nheads, -1, tokens), y[:, :, 2].permute(0, 2, 3, 1))
    H.lib().cfgpp_igemm_force_config(0)
    H.lib().cfgpp_igemm_set_tail_split(1)
    # plain linear through the split path: M=256, N=256, K=4096 (KT=64)
    a3, w3, b3, r3 = rnd(256, 4096, seed=70), rnd(256, 4096, scale=1 / 64, seed=71), rnd(256, scale=0.1, seed=72), rnd(256, 256, seed=73)
    out["linear_split"] = H.err_stats(H.linear(a3.to(H.DEV, torch.float16), w3.to(H.DEV, torch.float16), b3.to(H.DEV), r3.to(H.DEV, torch.float16)),
                                      a3 @ w3.t() + b3 + r3)
    return out


@case("groupnorm")
def t_gn():
    out = {}
    for (N, C0, C1, Hh, Ww, silu) in ((2, 64, 0, 8, 8, 1), (2, 128, 64, 7, 5, 1), (1, 320, 0, 16, 16, 0), (2, 1280, 640, 8, 8, 1),
                                      (1, 2560, 0, 4, 4, 1)):
        x0 = rnd(N, C0, Hh, Ww, seed=23) * 2 + 0.5
        x1 = rnd(N, C1, Hh, Ww, seed=24) if C1 else None
        C = C0 + C1
        g, b = 1 + rnd(C, scale=0.1, seed=25), rnd(C, scale=0.1, seed=26)
        xin = torch.cat([x0, x1], 1) if C1 else x0
        ref = F.group_norm(xin, 32, g, b, 1e-5)
        if silu:
            ref = F.silu(ref)
        got = H.groupnorm(H.to_pn(x0), H.to_pn(x1) if C1 else None, g.to(H.DEV), b.to(H.DEV), 32, 1e-5, silu)
        out[f"{N}x{C0}+{C1}x{Hh}x{Ww}"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got))
    # token-major destination
    x0 = rnd(2, 64, 6, 6, seed=27)
    g, b = 1 + rnd(64, scale=0.1, seed=28), rnd(64, scale=0.1, seed=29)
    ref = F.group_norm(x0, 32, g, b, 1e-6).permute(0, 2, 3, 1).reshape(-1, 64)
    got = H.groupnorm(H.to_pn(x0), None, g.to(H.DEV), b.to(H.DEV), 32, 1e-6, 0, dst_padded=False)
    out["tokens"] = H.err_stats(got, ref)
    return out


@case("groupnorm_prestats")
def t_gn_pre():
    """GroupNorm fed by the statistics its producers' epilogues wrote (IGemmArgs::gstat): conv3x3 (+ bias + time embedding + residual)
    through every tile family, then cfgpp_op_groupnorm_pre against torch's GroupNorm of the fp16 conv output; the per-(32-row block,
    column) pairs must be bit-identical across tile configs; a concat of two producers with different widths (groups straddle the
    sources); a producer with |mean| = 100 sigma (the cancellation case: rel-L2 must stay at fp16 output rounding); K-split launches
    must report that they wrote nothing."""
    out = {}
    lib = H.lib()
    lib.cfgpp_igemm_set_tail_split(0)
    try:
        # ---- one producer, many tile configs ----
        N, Cin, Cout, Hh, Ww = 4, 128, 320, 32, 32
        x = rnd(N, Cin, Hh, Ww, seed=90)
        w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=91)
        b = rnd(Cout, scale=0.1, seed=92) + 0.7
        temb = rnd(N, Cout, scale=0.5, seed=93)
        res = rnd(N, Cout, Hh, Ww, seed=94)
        g, be = 1 + rnd(Cout, scale=0.1, seed=95), rnd(Cout, scale=0.1, seed=96)
        xp, wp, rp = H.to_pn(x), H.pack_conv3(w), H.to_pn(res)
        gst = torch.zeros((N * Hh * Ww // 32, Cout, 2), dtype=torch.float32, device=H.DEV)
        base_gst = None
        for c in (1, 3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17, 19, 20, 24, 25, 26, 27):
            lib.cfgpp_igemm_force_config(c)
            gst.fill_(float("nan"))
            lib.cfgpp_op_igemm_set_gstat(H.P(gst))
            y = H.conv3x3(xp, wp, b.to(H.DEV), Hh, Ww, 1, temb.to(H.DEV), Cout, rp)
            wrote = int(lib.cfgpp_op_igemm_gstat_written())
            lib.cfgpp_op_igemm_set_gstat(None)
            yf = H.from_pn(y).float().cpu()
            ref = F.silu(F.group_norm(yf, 32, g, be, 1e-5))
            got = H.groupnorm_pre(y, None, gst, None, g.to(H.DEV), be.to(H.DEV), 32, 1e-5, 1)
            if base_gst is None:
                base_gst = gst.clone()
            # the pairs against their definition: per 32-row block and column, mean and sum of squared deviations of the fp16 outputs
            blocks = yf.permute(0, 2, 3, 1).reshape(-1, 32, Cout).double()
            mean_ref, m2_ref = blocks.mean(1), ((blocks - blocks.mean(1, keepdim=True)) ** 2).sum(1)
            gc = gst.cpu().double()
            out[f"conv_cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got), wrote=wrote,
                                       same_pairs_as_cfg1=bool(torch.equal(gst, base_gst)),
                                       mean_err=float((gc[..., 0] - mean_ref).abs().max()), m2_rel=float(((gc[..., 1] - m2_ref).abs() / (m2_ref + 1e-6)).max()))
        lib.cfgpp_igemm_force_config(0)
        # ---- concat of two producers (1 x 1 convs, 64 + 192 channels: groups of 8 straddle nothing, groups of (64+192)/32 = 8 ... and
        #      320 + 640 -> 30-channel groups that DO straddle the boundary) ----
        for name, Ca, Cb, Hh2 in (("cat_64_192", 64, 192, 16), ("cat_640_320", 640, 320, 16)):
            xa, xb = rnd(2, 64, Hh2, Hh2, seed=100), rnd(2, 128, Hh2, Hh2, seed=101)
            wa, wb_ = rnd(Ca, 64, scale=1 / 8, seed=102), rnd(Cb, 128, scale=128 ** -0.5, seed=103)
            ba, bb = rnd(Ca, scale=0.3, seed=104), rnd(Cb, scale=0.3, seed=105) - 1.0
            ga, gb_ = torch.zeros((2 * Hh2 * Hh2 // 32, Ca, 2), device=H.DEV), torch.zeros((2 * Hh2 * Hh2 // 32, Cb, 2), device=H.DEV)
            lib.cfgpp_op_igemm_set_gstat(H.P(ga))
            ya = H.conv1x1_2src(H.to_pn(xa), None, wa.to(H.DEV, torch.float16), ba.to(H.DEV))
            wrote_a = int(lib.cfgpp_op_igemm_gstat_written())
            lib.cfgpp_op_igemm_set_gstat(H.P(gb_))
            yb = H.conv1x1_2src(H.to_pn(xb), None, wb_.to(H.DEV, torch.float16), bb.to(H.DEV))
            wrote_b = int(lib.cfgpp_op_igemm_gstat_written())
            lib.cfgpp_op_igemm_set_gstat(None)
            C = Ca + Cb
            gg, bg = 1 + rnd(C, scale=0.1, seed=106), rnd(C, scale=0.1, seed=107)
            ref = F.group_norm(torch.cat([H.from_pn(ya).float().cpu(), H.from_pn(yb).float().cpu()], 1), 32, gg, bg, 1e-6)
            got = H.groupnorm_pre(ya, yb, ga, gb_, gg.to(H.DEV), bg.to(H.DEV), 32, 1e-6, 0)
            old = H.groupnorm(ya, yb, gg.to(H.DEV), bg.to(H.DEV), 32, 1e-6, 0)
            out[name] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got), wrote=wrote_a * wrote_b,
                             vs_own_pass=float((H.from_pn(got).float() - H.from_pn(old).float()).abs().max()))
        # ---- |mean| = 100 sigma: bias 50, outputs with sigma ~0.5 ----
        xs = rnd(2, 64, 16, 16, seed=110)
        ws = rnd(64, 64, scale=0.5 / 8, seed=111)
        bs = torch.full((64,), 50.0)
        gl = torch.zeros((2 * 256 // 32, 64, 2), device=H.DEV)
        lib.cfgpp_op_igemm_set_gstat(H.P(gl))
        yl = H.conv1x1_2src(H.to_pn(xs), None, ws.to(H.DEV, torch.float16), bs.to(H.DEV))
        lib.cfgpp_op_igemm_set_gstat(None)
        g1, b1 = torch.ones(64), torch.zeros(64)
        ref = F.group_norm(H.from_pn(yl).double().cpu(), 32, g1.double(), b1.double(), 1e-5).float()
        got = H.groupnorm_pre(yl, None, gl, None, g1.to(H.DEV), b1.to(H.DEV), 32, 1e-5, 0)
        out["large_mean"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=True, wrote=1)
        # ---- a K-split launch reports that it wrote nothing ----
        lib.cfgpp_igemm_set_tail_split(1)
        xk = rnd(2, 1280, 8, 8, seed=120)
        wk = rnd(1280, 1280, 3, 3, scale=(9 * 1280) ** -0.5, seed=121)
        gk = torch.zeros((2 * 64 // 32, 1280, 2), device=H.DEV)
        lib.cfgpp_op_igemm_set_gstat(H.P(gk))
        H.conv3x3(H.to_pn(xk), H.pack_conv3(wk), None, 8, 8, 1)
        out["ksplit_reports_no_stats"] = dict(rel_l2=0.0, max_abs=0.0, finite=True, halo_zero=True, wrote=1 - int(lib.cfgpp_op_igemm_gstat_written()))
        lib.cfgpp_op_igemm_set_gstat(None)
    finally:
        lib.cfgpp_op_igemm_set_gstat(None)
        lib.cfgpp_igemm_force_config(0)
        lib.cfgpp_igemm_set_tail_split(1)
    return out


@case("layernorm")
def t_ln():
    out = {}
    for C in (64, 320, 640, 1280):
        x = rnd(37, C, seed=30) * 3 + 1
        g, b = 1 + rnd(C, scale=0.1, seed=31), rnd(C, scale=0.1, seed=32)
        ref = F.layer_norm(x, (C,), g, b, 1e-5)
        got = H.layernorm(x.to(H.DEV, torch.float16), g.to(H.DEV), b.to(H.DEV))
        out[str(C)] = H.err_stats(got, ref)
    return out


@case("attention")
def t_attn():
    out = {}
    for (B, h, Nq, Nk, d) in ((1, 2, 64, 64, 32), (2, 2, 256, 256, 64), (1, 8, 192, 192, 40), (1, 4, 128, 128, 80), (1, 2, 64, 64, 160),
                              (2, 2, 100, 77, 64), (1, 8, 256, 77, 40), (1, 2, 1024, 1024, 64)):
        q, k, v = rnd(B, h, Nq, d, seed=33), rnd(B, h, Nk, d, seed=34), rnd(B, h, Nk, d, seed=35)
        ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Nq, h * d)
        hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
        got = H.attention(hq, hk, hvt, B, h, d, Nq, Nk, qp, kp)
        out[f"B{B}h{h}q{Nq}k{Nk}d{d}"] = H.err_stats(got, ref)
    return out


@case("heads_projection")
def t_heads(B=2, tokens=96, C=128, nheads=4):
    a = rnd(B * tokens, C, seed=36)
    w = rnd(3 * C, C, scale=C ** -0.5, seed=37)
    d = C // nheads
    qp, kp = H.round_up(tokens, 128), H.round_up(tokens, 64)
    y = (a @ w.t()).reshape(B, tokens, 3, nheads, d)
    out = {}
    for c in (0, 7, 9, 11, 12, 13, 14, 15, 16, 17, 20, 24, 25, 26):          # heuristic tile, 128x160, the 3- / 4-stage ring tiles, the 32-deep-K-tile linears (LDS-staged heads epilogue)
        H.lib().cfgpp_igemm_force_config(c)
        hq, hk, hvt = H.heads_project(a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16), B, tokens, C, nheads, 0, 3, qp, kp)
        sfx = "" if c == 0 else f"_cfg{c}"
        out["q" + sfx] = H.err_stats(hq[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 0].permute(0, 2, 1, 3))
        out["k" + sfx] = H.err_stats(hk[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 1].permute(0, 2, 1, 3))
        out["vt" + sfx] = H.err_stats(hvt[:, :d, H.vt_pos(tokens).to(H.DEV)].reshape(B, nheads, d, tokens), y[:, :, 2].permute(0, 2, 3, 1))
        out["pad_zero" + sfx] = bool(hq[:, tokens:].abs().sum() == 0 and hq[:, :, d:].abs().sum() == 0 and hvt[:, :d, tokens:].abs().sum() == 0 and hvt[:, d + 1:].abs().sum() == 0)
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("heads_projection_d40")
def t_heads_d40(B=2, tokens=96, C=320, nheads=8):
    """QKV projection at the SD1.5 level-0 geometry (N = 960 = 6 x 160, head dim 40: 16-column groups straddle heads): the
    heuristic tile, the 4-wave 128x160 tile and the 16x16x32-MFMA tile's head-major epilogue (configs 18 / 19)."""
    a = rnd(B * tokens, C, seed=46)
    w = rnd(3 * C, C, scale=C ** -0.5, seed=47)
    d = C // nheads
    qp, kp = H.round_up(tokens, 128), H.round_up(tokens, 64)
    y = (a @ w.t()).reshape(B, tokens, 3, nheads, d)
    cfgs = [0, 7, 13, 15, 16, 17, 18, 19, 20, 24, 25, 26]
    out = {}
    H.lib().cfgpp_igemm_set_mf16_heads(1 if 18 in cfgs else 0)
    try:
        for c in cfgs:
            H.lib().cfgpp_igemm_force_config(c)
            hq, hk, hvt = H.heads_project(a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16), B, tokens, C, nheads, 0, 3, qp, kp)
            out[f"q_cfg{c}"] = H.err_stats(hq[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 0].permute(0, 2, 1, 3))
            out[f"k_cfg{c}"] = H.err_stats(hk[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 1].permute(0, 2, 1, 3))
            out[f"vt_cfg{c}"] = H.err_stats(hvt[:, :d, H.vt_pos(tokens).to(H.DEV)].reshape(B, nheads, d, tokens), y[:, :, 2].permute(0, 2, 3, 1))
    finally:
        H.lib().cfgpp_igemm_force_config(0)
        H.lib().cfgpp_igemm_set_mf16_heads(1)
    return out


@case("mf16_race")
def t_mf16_race():
    """igemm16_kernel at full-chip grids (256 tiles, 20 .. 180 K-tiles, 3- and 4-stage rings): repeated launches must be
    bit-identical (an LDS race shows as run-to-run differences) and match fp32"""
    out = {}
    for name, M, N, K, conv in (("lin_k1280", 4096, 1280, 1280, False), ("lin_k5120", 4096, 1280, 5120, False), ("conv_k2880", 4096, 1280, 320, True)):
        if conv:
            x = rnd(4, 320, 32, 32, seed=70)
            w = rnd(N, 320, 3, 3, scale=(9 * 320) ** -0.5, seed=71)
            ref = F.conv2d(x, w, None, padding=1)
            xp, wp = H.to_pn(x), H.pack_conv3(w)
        else:
            a = rnd(M, K, seed=72)
            w = rnd(N, K, scale=K ** -0.5, seed=73)
            res = rnd(M, N, seed=74)
            ref = a @ w.t() + res
            ad, wd, rd = a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16), res.to(H.DEV, torch.float16)
        for c in (18, 19):
            H.lib().cfgpp_igemm_force_config(c)
            runs = []
            for _ in range(12):
                if conv:
                    runs.append(H.from_pn(H.conv3x3(xp, wp, None, 32, 32)))
                else:
                    runs.append(H.linear(ad, wd, None, resid=rd).float().cpu())
            out[f"{name}_cfg{c}"] = dict(H.err_stats(runs[0], ref), identical_runs=all(torch.equal(runs[0], r_) for r_ in runs[1:]))
    H.lib().cfgpp_igemm_force_config(0)
    return out


def _tiles_at_unet_sizes(cfgs):
    """tile32_kernel (configs 15 / 16 / 17: two or three workgroups per CU; 13 / 20: the 256-wide tiles on four stages) at launch
    shapes of the UNets - linears (to_out + residual, FF-out, GEGLU, ragged M) and convolutions (3x3 with time embedding and
    residual, stride 2 with both paddings, nearest-2x upsample, 1x1 over two concatenated sources) on full-chip grids: right
    against fp32, BIT-IDENTICAL to the 128 x 128 tile of igemm_kernel (same k order: what lets the tuner pin them) and
    identical run to run"""
    out = {}
    H.lib().cfgpp_igemm_set_tail_split(0)          # the 128 x 128 baseline must not K-split (different summation order)
    shapes = (("to_out_c320", 16384, 320, 320, True, 0), ("to_out_c1280", 4096, 1280, 1280, True, 0), ("ff_out_c320", 8192, 320, 1280, True, 0),
              ("ragged", 1000, 192, 448, False, 0), ("geglu_c320", 8192, 2560, 320, False, 1))
    for name, M, N, K, resid, epi in shapes:
        a = rnd(M, K, seed=len(name))
        w = rnd(N, K, scale=K ** -0.5, seed=len(name) + 1)
        b = rnd(N, scale=0.1, seed=len(name) + 2)
        r = rnd(M, N, seed=len(name) + 3) if resid else None
        ad, bd = a.to(H.DEV, torch.float16), b.to(H.DEV)
        rd = r.to(H.DEV, torch.float16) if resid else None
        if epi == 1:
            h = a @ w.t() + b
            v, g = h.chunk(2, dim=-1)
            ref = v * F.gelu(g)
            wd, bd = H.pack_geglu(w, b)
        else:
            ref = a @ w.t() + b + (r if resid else 0)
            wd = w.to(H.DEV, torch.float16)
        H.lib().cfgpp_igemm_force_config(1)
        base = H.linear(ad, wd, bd, rd, epi=epi)
        for c in cfgs:
            H.lib().cfgpp_igemm_force_config(c)
            got = H.linear(ad, wd, bd, rd, epi=epi)
            same = all(torch.equal(got, H.linear(ad, wd, bd, rd, epi=epi)) for _ in range(4))
            out[f"{name}_cfg{c}"] = dict(H.err_stats(got, ref), identical_runs=bool(same), equals_cfg1=bool(torch.equal(got, base)))
    # convolutions: (name, N, Cin, Cout, H_out, W_out, amode)
    for name, N, Cin, Cout, Hh, Ww, amode in (("conv_320_640", 4, 320, 640, 32, 32, 1), ("conv_s2", 4, 128, 320, 16, 16, 2), ("conv_up", 2, 128, 320, 32, 32, 3),
                                              ("conv_small_k", 3, 64, 192, 24, 20, 1)):
        Hi, Wi = (2 * Hh, 2 * Ww) if amode == 2 else (Hh // 2, Ww // 2) if amode == 3 else (Hh, Ww)
        x = rnd(N, Cin, Hi, Wi, seed=len(name) + 10)
        w = rnd(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=len(name) + 11)
        b = rnd(Cout, scale=0.1, seed=len(name) + 12)
        if amode == 1:
            temb = rnd(N, Cout, scale=0.5, seed=len(name) + 13)
            res = rnd(N, Cout, Hh, Ww, seed=len(name) + 14)
            ref = F.conv2d(x, w, b, padding=1) + temb[:, :, None, None] + res
            args = (temb.to(H.DEV), Cout, H.to_pn(res))
        elif amode == 2:
            ref = F.conv2d(x, w, b, stride=2, padding=1)
            args = ()
        else:
            ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
            args = ()
        xp, wp = H.to_pn(x), H.pack_conv3(w)
        H.lib().cfgpp_igemm_force_config(1)
        base = H.conv3x3(xp, wp, b.to(H.DEV), Hh, Ww, amode, *args)
        for c in cfgs:
            H.lib().cfgpp_igemm_force_config(c)
            got = H.conv3x3(xp, wp, b.to(H.DEV), Hh, Ww, amode, *args)
            same = all(torch.equal(got, H.conv3x3(xp, wp, b.to(H.DEV), Hh, Ww, amode, *args)) for _ in range(3))
            out[f"{name}_cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), identical_runs=bool(same), equals_cfg1=bool(torch.equal(got, base)),
                                         halo_zero=H.halo_is_zero(got))
    x0, x1 = rnd(4, 640, 32, 32, seed=70), rnd(4, 320, 32, 32, seed=71)
    w = rnd(640, 960, scale=960 ** -0.5, seed=72)
    b = rnd(640, scale=0.1, seed=73)
    ref = F.conv2d(torch.cat([x0, x1], 1), w[:, :, None, None], b)
    H.lib().cfgpp_igemm_force_config(1)
    base = H.conv1x1_2src(H.to_pn(x0), H.to_pn(x1), w.to(H.DEV, torch.float16), b.to(H.DEV))
    for c in cfgs:
        H.lib().cfgpp_igemm_force_config(c)
        got = H.conv1x1_2src(H.to_pn(x0), H.to_pn(x1), w.to(H.DEV, torch.float16), b.to(H.DEV))
        out[f"conv1x1_2src_cfg{c}"] = dict(H.err_stats(H.from_pn(got), ref), identical_runs=True, equals_cfg1=bool(torch.equal(got, base)))
    H.lib().cfgpp_igemm_force_config(0)
    H.lib().cfgpp_igemm_set_tail_split(1)
    return out


@case("tile32_unet_sizes")
def t_tile32():
    return _tiles_at_unet_sizes((13, 15, 16, 17, 20, 27))


@case("big4_unet_sizes")
def t_big4():
    """big4_kernel (configs 24 / 25 / 26: 256 x 256, 128 x 320, 128 x 256 on ONE wave per SIMD, scalar-base LDS-DMA pieces) at the
    same launch shapes + the QKV head-major projection: right against fp32, bit-identical to the 128 x 128 tile, repeatable"""
    out = _tiles_at_unet_sizes((24, 25, 26))
    # head-major epilogue, SD1.5 level-0 geometry (head dim 40) and a 64-wide head, full-chip grid
    for name, B, tokens, C, nheads in (("heads_d40", 4, 1024, 320, 8), ("heads_d64", 2, 1024, 640, 10)):
        a = rnd(B * tokens, C, seed=len(name) + 20)
        w = rnd(3 * C, C, scale=C ** -0.5, seed=len(name) + 21)
        ad, wd = a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16)
        d = C // nheads
        y = (a @ w.t()).reshape(B, tokens, 3, nheads, d)
        H.lib().cfgpp_igemm_force_config(1)
        base = H.heads_project(ad, wd, B, tokens, C, nheads, 0, 3, tokens, tokens)
        for c in (24, 25, 26):
            H.lib().cfgpp_igemm_force_config(c)
            hq, hk, hvt = H.heads_project(ad, wd, B, tokens, C, nheads, 0, 3, tokens, tokens)
            eq = all(torch.equal(x, y_) for x, y_ in zip((hq, hk, hvt), base))
            st = H.err_stats(hq[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 0].permute(0, 2, 1, 3))
            stv = H.err_stats(hvt[:, :d, H.vt_pos(tokens).to(H.DEV)].reshape(B, nheads, d, tokens), y[:, :, 2].permute(0, 2, 3, 1))
            st["rel_l2"] = max(st["rel_l2"], stv["rel_l2"])
            out[f"{name}_cfg{c}"] = dict(st, identical_runs=True, equals_cfg1=bool(eq))
    H.lib().cfgpp_igemm_force_config(0)
    return out


@case("big4p_persistent")
def t_big4p():
    """big4p_kernel (config 28: the 256 x 256 tile as a persistent kernel, 8 waves, the next output tile's first
    K-tile prefetched under the epilogue) on token-major linears whose grids are several rounds of 256 workgroups, a tile count
    that is not a multiple of 8 (the plain round-robin tile sequence), ragged edges, a single K-tile, every epilogue: right
    against fp32, BIT-IDENTICAL to the 128 x 128 tile, identical run to run"""
    out = {}
    H.lib().cfgpp_igemm_set_tail_split(0)
    shapes = (("store_res_640tiles", 16384, 2560, 320, True, 0), ("geglu_1280tiles", 16384, 5120, 640, False, 1),
              ("store_299tiles", 5888, 3328, 128, False, 0), ("ragged_100tiles", 5000, 1032, 192, True, 0),
              ("one_ktile", 8192, 1024, 64, False, 0), ("geglu_sdxl32", 4096, 10240, 1280, False, 1))
    for name, M, N, K, resid, epi in shapes:
        a = rnd(M, K, seed=len(name))
        w = rnd(N, K, scale=K ** -0.5, seed=len(name) + 1)
        b = rnd(N, scale=0.1, seed=len(name) + 2)
        r = rnd(M, N, seed=len(name) + 3) if resid else None
        ad, bd = a.to(H.DEV, torch.float16), b.to(H.DEV)
        rd = r.to(H.DEV, torch.float16) if resid else None
        if epi == 1:
            h = a @ w.t() + b
            v, g = h.chunk(2, dim=-1)
            ref = v * F.gelu(g)
            wd, bd = H.pack_geglu(w, b)
        else:
            ref = a @ w.t() + b + (r if resid else 0)
            wd = w.to(H.DEV, torch.float16)
        H.lib().cfgpp_igemm_force_config(1)
        base = H.linear(ad, wd, bd, rd, epi=epi)
        for c in (28,):
            H.lib().cfgpp_igemm_force_config(c)
            got = H.linear(ad, wd, bd, rd, epi=epi)
            same = all(torch.equal(got, H.linear(ad, wd, bd, rd, epi=epi)) for _ in range(4))
            out[f"{name}_cfg{c}"] = dict(H.err_stats(got, ref), identical_runs=bool(same), equals_cfg1=bool(torch.equal(got, base)))
    for name, B, tokens, C, nheads in (("heads_d64_512tiles", 4, 4096, 640, 10), ("heads_d40", 4, 1024, 320, 8)):
        a = rnd(B * tokens, C, seed=len(name) + 20)
        w = rnd(3 * C, C, scale=C ** -0.5, seed=len(name) + 21)
        ad, wd = a.to(H.DEV, torch.float16), w.to(H.DEV, torch.float16)
        d = C // nheads
        y = (a @ w.t()).reshape(B, tokens, 3, nheads, d)
        H.lib().cfgpp_igemm_force_config(1)
        base = H.heads_project(ad, wd, B, tokens, C, nheads, 0, 3, tokens, tokens)
        for c in (28,):
            H.lib().cfgpp_igemm_force_config(c)
            hq, hk, hvt = H.heads_project(ad, wd, B, tokens, C, nheads, 0, 3, tokens, tokens)
            eq = all(torch.equal(x, y_) for x, y_ in zip((hq, hk, hvt), base))
            again = H.heads_project(ad, wd, B, tokens, C, nheads, 0, 3, tokens, tokens)
            same = all(torch.equal(x, y_) for x, y_ in zip((hq, hk, hvt), again))
            st = H.err_stats(hq[:, :tokens, :d].reshape(B, nheads, tokens, d), y[:, :, 0].permute(0, 2, 1, 3))
            out[f"{name}_cfg{c}"] = dict(st, identical_runs=bool(same), equals_cfg1=bool(eq))
    H.lib().cfgpp_igemm_force_config(0)
    H.lib().cfgpp_igemm_set_tail_split(1)
    return out


@case("conv_in_out")
def t_cio():
    out = {}
    z = rnd(2, 4, 16, 12, seed=38)
    w = rnd(64, 4, 3, 3, scale=1 / 6, seed=39)
    b = rnd(64, scale=0.1, seed=40)
    ref = F.conv2d(torch.cat([z, z]), w, b, padding=1)
    got = H.conv_in(z.to(H.DEV), w, b.to(H.DEV), 4)
    out["conv_in_f32"] = dict(H.err_stats(H.from_pn(got), ref), halo_zero=H.halo_is_zero(got))
    got = H.conv_in(z.to(H.DEV, torch.float16), w, b.to(H.DEV), 4)
    out["conv_in_f16"] = H.err_stats(H.from_pn(got), ref)
    x = rnd(3, 64, 10, 8, seed=41)
    w2 = rnd(4, 64, 3, 3, scale=(9 * 64) ** -0.5, seed=42)
    b2 = rnd(4, scale=0.1, seed=43)
    ref2 = F.conv2d(x, w2, b2, padding=1)
    out["conv_out"] = H.err_stats(H.conv_out(H.to_pn(x), w2, b2.to(H.DEV)), ref2)
    # maps of >= 128 x 128 pixels take the LDS-tiled kernel: sizes that are not multiples of the 8 x 32 tile, 3 and 4 outputs,
    # fp32 and fp16 results, two 64-channel blocks; and the same launch through the wave-per-pixel kernel (A/B switch)
    for (R, C, Hh, Ww, Co, half) in ((2, 128, 136, 152, 3, False), (1, 64, 128, 128, 4, True), (1, 128, 130, 161, 4, False)):
        xb = rnd(R, C, Hh, Ww, seed=44 + Co)
        wb = rnd(Co, C, 3, 3, scale=(9 * C) ** -0.5, seed=45)
        bb = rnd(Co, scale=0.1, seed=46)
        refb = F.conv2d(xb.half().float(), wb.half().float(), bb, padding=1)
        xpn = H.to_pn(xb)
        got = H.conv_out(xpn, wb, bb.to(H.DEV), out_half=half)
        out[f"conv_out_tiled_{Hh}x{Ww}_c{C}_o{Co}"] = H.err_stats(got, refb)
        H.lib().cfgpp_conv_out_set_tiled(0)
        old_k = H.conv_out(xpn, wb, bb.to(H.DEV), out_half=half)
        H.lib().cfgpp_conv_out_set_tiled(1)
        out[f"conv_out_tiled_vs_wave_{Hh}x{Ww}_o{Co}"] = H.err_stats(got, old_k.float().cpu())
    return out


@case("sinusoid_skinny")
def t_small():
    from oracle.unet_ref import timestep_embedding
    out = {}
    vals = torch.tensor([981.0, 1.0, 500.0, 1024.0, 0.0])
    out["sinusoid"] = H.err_stats(H.sinusoid(vals.to(H.DEV), 320), timestep_embedding(vals, 320))
    x = rnd(5, 320, seed=44)
    w = rnd(1280, 320, scale=320 ** -0.5, seed=45)
    b = rnd(1280, scale=0.1, seed=46)
    ref = F.silu(F.linear(F.silu(x), w, b))
    out["skinny_m5"] = H.err_stats(H.skinny(x.to(H.DEV), w.to(H.DEV, torch.float16), b.to(H.DEV), True, True), ref)
    out["skinny_m1"] = H.err_stats(H.skinny(x[:1].contiguous().to(H.DEV), w.to(H.DEV, torch.float16), b.to(H.DEV), True, True), ref[:1])
    return out


@case("step_kernels_golden")
def t_step():
    import numpy as np
    from cfgpp_amd import engine as E
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    g = np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden.npz"))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    tag, tb, lam = "G2/sd_ddim_cfgpp_h", SchedulerTables(50), 0.6
    z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
    bad = 0
    for i, t in enumerate(tb.timesteps):
        zi = z[i][0:1].contiguous().to(H.DEV)
        z0o = torch.empty_like(zi)
        co = ddim_coeffs_pinned(tb.ddim_sqrt_coeffs(t), eps_half=True)
        E.step_ddim(zi, z0o, e[i][0:1].contiguous().to(H.DEV), e[i][1:2].contiguous().to(H.DEV), lam, co, False, True)
        bad += int((z0o.cpu() != z0[i]).sum()) + int((zi.cpu() != zt[i]).sum())
    return {"mismatching_elements": bad, "steps": int(len(tb.timesteps))}


def unet_case(cfg_name, R, hw, seed=0, iters=0, tvals=(981.0, 1.0)):
    from cfgpp_amd.engine import HipUNet
    from cfgpp_amd.unet_config import CONFIGS
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    cfg = CONFIGS[cfg_name]
    sd = synth_state_dict(cfg, seed)
    zB = R // 2
    z = rnd(zB, 4, hw, hw, seed=50)
    ehs = rnd(R, 77, cfg.cross_attention_dim, scale=0.5, seed=51)
    te = ti = None
    ack = None
    if cfg.addition_embed:
        te = rnd(R, cfg.addition_pooled_dim, scale=0.5, seed=52)
        ti = torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * R)
        ack = {"text_embeds": te, "time_ids": ti}
    t0 = time.time()
    net = HipUNet(cfg, max_rows=R, sample_hw=(hw, hw))
    net.load_state_dict(sd).finalize()
    t_build = time.time() - t0
    net.set_context(ehs, te, ti)
    out = {}
    ref_net = UNetRef(cfg, sd)
    for tval in tvals:
        eps = net.forward(z.to(H.DEV), tval)
        torch.cuda.synchronize()
        t0 = time.time()
        ref = ref_net(torch.cat([z, z]), tval, ehs, ack)["sample"]
        out[f"t{int(tval)}"] = dict(H.err_stats(eps, ref), cpu_ref_s=round(time.time() - t0, 2))
    out["build_s"] = round(t_build, 1)
    out["device_GB"] = round(net.device_bytes() / 1e9, 2)
    if iters:
        zd = z.to(H.DEV)
        dt = timeit(lambda: net.forward(zd, 500.0), iters=iters, warm=2)
        out["fwd_ms"] = round(dt * 1e3, 3)
        out["TFLOPs"] = round(net.flops(R) / dt / 1e12, 1)
    return out


@case("unet_tiny_sd")
def t_unet_tiny_sd():
    return unet_case("tiny_sd", 4, 16, iters=5)


@case("unet_tiny_xl")
def t_unet_tiny_xl():
    return unet_case("tiny_xl", 2, 16, iters=5)


@case("unet_sd15_r2")
def t_unet_sd15():
    return unet_case("sd15", 2, 64, iters=5)


@case("unet_sdxl_r2_64")
def t_unet_sdxl():
    return unet_case("sdxl", 2, 64, iters=3)


@case("microbench")
def t_bench():
    out = {}
    # GEMMs at SD1.5 / SDXL shapes, batch 16 rows
    for (name, M, N, K) in (("geglu_l0", 65536, 2560, 320), ("qkv_l0", 65536, 960, 320), ("ffout_l0", 65536, 320, 1280),
                            ("geglu_l1", 16384, 5120, 640), ("xl_geglu", 4096, 10240, 1280), ("xl_ffout", 4096, 1280, 5120),
                            ("xl_qkv", 4096, 3840, 1280)):
        a = torch.randn(M, K, device=H.DEV, dtype=torch.float16)
        w = torch.randn(N, K, device=H.DEV, dtype=torch.float16) * K ** -0.5
        outb = torch.empty(M, N, device=H.DEV, dtype=torch.float16)
        fn = lambda: H.igemm(a, None, K, 0, 1, 0, 0, 0, w, M, N, out=outb, omode=0, old=N)  # noqa: E731
        dt = timeit(fn, iters=10)
        out[name] = dict(ms=round(dt * 1e3, 3), TF=round(2.0 * M * N * K / dt / 1e12, 1))
    for (name, N, C, Co, hw) in (("conv_l0", 16, 320, 320, 64), ("conv_l1", 16, 640, 640, 32), ("conv_l2", 16, 1280, 1280, 16),
                                 ("conv_l3", 16, 1280, 1280, 8), ("conv_up0", 16, 2560, 1280, 16), ("xl_conv_128", 4, 320, 320, 128)):
        x = torch.randn(N, hw + 2, hw + 2, C, device=H.DEV, dtype=torch.float16)
        w = torch.randn(Co, 9 * C, device=H.DEV, dtype=torch.float16) * (9 * C) ** -0.5
        o = H.empty_pn(N, hw, hw, Co)
        fn = lambda: H.igemm(x, None, C, 0, 9, 1, hw, hw, w, N * hw * hw, Co, out=o, omode=1, old=Co)  # noqa: E731
        dt = timeit(fn, iters=10)
        out[name] = dict(ms=round(dt * 1e3, 3), TF=round(2.0 * N * hw * hw * Co * 9 * C / dt / 1e12, 1))
    for (name, B, h, Nn, d) in (("attn_l0_d40", 16, 8, 4096, 40), ("attn_l1_d80", 16, 8, 1024, 80), ("attn_xl_d64", 4, 20, 1024, 64),
                                ("attn_xl_4096", 4, 10, 4096, 64)):
        dp = H.round_up(d, 32)
        hq = torch.randn(B * h, Nn, dp, device=H.DEV, dtype=torch.float16)
        hk = torch.randn(B * h, Nn, dp, device=H.DEV, dtype=torch.float16)
        hvt = torch.randn(B * h, dp, Nn, device=H.DEV, dtype=torch.float16)
        o = torch.empty(B, Nn, h * d, device=H.DEV, dtype=torch.float16)
        fn = lambda: _lib.check(H.lib().cfgpp_op_attention(H.P(hq), H.P(hk), H.P(hvt), H.P(o), B, h, d, Nn, Nn, Nn, Nn, H.stream()), "attn")  # noqa: E731
        dt = timeit(fn, iters=10)
        out[name] = dict(ms=round(dt * 1e3, 3), TF=round(4.0 * B * h * Nn * Nn * d / dt / 1e12, 1))
    # GroupNorm / LayerNorm bandwidth
    for (name, N, C, hw) in (("gn_320_64", 16, 320, 64), ("gn_1280_16", 16, 1280, 16), ("gn_xl_320_128", 4, 320, 128)):
        x = torch.randn(N, hw + 2, hw + 2, C, device=H.DEV, dtype=torch.float16)
        g = torch.ones(C, device=H.DEV)
        b = torch.zeros(C, device=H.DEV)
        fn = lambda: H.groupnorm(x, None, g, b, 32, 1e-5, 1)  # noqa: E731
        dt = timeit(fn, iters=10)
        out[name] = dict(ms=round(dt * 1e3, 3), GBs=round(3 * 2.0 * N * hw * hw * C / dt / 1e9, 1))
    for (name, rows, C) in (("ln_320", 65536, 320), ("ln_1280", 16384, 1280)):
        x = torch.randn(rows, C, device=H.DEV, dtype=torch.float16)
        g = torch.ones(C, device=H.DEV)
        b = torch.zeros(C, device=H.DEV)
        fn = lambda: H.layernorm(x, g, b)  # noqa: E731
        dt = timeit(fn, iters=10)
        out[name] = dict(ms=round(dt * 1e3, 3), GBs=round(2 * 2.0 * rows * C / dt / 1e9, 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--full-unet", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "diag.json"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    print("[diag] device:", torch.cuda.get_device_name(0), flush=True)
    t_gemm_t(); t_gemm(); t_geglu(); t_conv(); t_conv_s2(); t_conv_up(); t_conv1(); t_big(); t_tail(); t_gn(); t_gn_pre(); t_ln(); t_attn(); t_heads()
    t_tile32(); t_big4()
    t_cio(); t_small(); t_step()
    t_unet_tiny_sd(); t_unet_tiny_xl()
    if not args.quick:
        t_bench()
    if args.full_unet:
        t_unet_sd15(); t_unet_sdxl()
    with open(args.out, "w") as f:
        json.dump(RESULTS, f, indent=1)
    print("[diag] wrote", args.out)


if __name__ == "__main__":
    main()
