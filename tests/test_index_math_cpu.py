"""CPU checks of the two index-math claims the round-3 epilogues rest on (cfgpp_amd/csrc/igemm_kernel.hip):

* ``qdiv``: floor(m / d) as one multiply by an fp32 reciprocal plus a one-step correction is EXACT for quotients below
  2^20, even when the hardware reciprocal (v_rcp_f32, 1 ulp) is off by two ulps;
* ``head_col`` / ``head_step``: (part, head, offset in the head) of a column computed once per aligned column group plus
  at most one conditional wrap equals the per-column ``n / part_width``, ``(n % part_width) / head_dim``,
  ``(n % part_width) % head_dim`` the epilogues used to evaluate per 16-byte piece.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _qdiv(m, d, ulps):
    inv = np.float32(1.0) / np.float32(d)
    for _ in range(abs(ulps)):
        inv = np.nextafter(inv, np.float32(np.inf if ulps > 0 else -np.inf))
    q = (m.astype(np.float32) * inv).astype(np.int64)          # v_cvt_f32_i32, v_mul_f32, v_cvt_i32_f32 (truncation)
    r = m - q * d
    return q + (r >= d).astype(np.int64) - (r < 0).astype(np.int64)


@pytest.mark.parametrize("ulps", [-2, -1, 0, 1, 2])
def test_reciprocal_division_is_exact_for_small_quotients(ulps):
    rng = np.random.default_rng(7 + ulps)
    # divisors: H*W of every level (and odd ones), image widths, tile-walk divisors
    divisors = [1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 16, 20, 24, 25, 36, 48, 63, 64, 65, 77, 96, 100, 128, 255, 256, 257, 1000, 1024,
                4096, 4097, 16384, 65536, 262144, 1048576, 1023 * 1025]
    for d in divisors:
        qmax = min(2 ** 20, (2 ** 31 - 1) // d)
        q = rng.integers(0, qmax, 60000)
        r = rng.integers(0, d, 60000)
        r[:20000] = 0                                           # exact multiples and the value just below them
        r[20000:40000] = d - 1
        m = q * d + r
        m = m[m < 2 ** 31 - 1]
        assert np.array_equal(_qdiv(m, d, ulps), m // d), d


def _head_step(dd0, head0, off, head_dim):
    dd, head = dd0 + off, head0
    if head_dim >= 32:
        if dd >= head_dim:
            dd -= head_dim
            head += 1
    else:
        head += dd // head_dim
        dd = dd % head_dim
    return head, dd


@pytest.mark.parametrize("group", [16, 32])
@pytest.mark.parametrize("head_dim", [8, 16, 40, 64, 80, 160])
def test_head_major_addressing_without_per_piece_divisions(group, head_dim):
    for heads in (1, 2, 5, 8, 20):
        part_width = heads * head_dim
        if part_width % group:
            continue                                            # the staged epilogues require aligned parts
        N = 3 * part_width
        for ng in range(0, N, group):                           # first column of an aligned group (wave-uniform)
            pr = ng // part_width
            cn0 = ng - pr * part_width
            head0, dd0 = cn0 // head_dim, cn0 % head_dim
            for off in range(group):
                n = ng + off
                cn = n % part_width
                assert n // part_width == pr                    # a group lies inside one part
                assert _head_step(dd0, head0, off, head_dim) == (cn // head_dim, cn % head_dim), (ng, off)


def test_packed_gelu_polynomial_constants():
    """csrc/common.h gelu_erf_pk: the shipped constants, evaluated the way the kernel evaluates them (fp32, clamp, centred
    variable, Horner with one rounding per step), against x * Phi(x) with scipy's erf."""
    import re

    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(ROOT, "cfgpp_amd", "csrc", "common.h")).read()
    c = np.float32(re.search(r"#define CFGPP_GELU_C ([0-9.]+)f", src).group(1))
    body = re.search(r"#define CFGPP_GELU_POLY \{(.*?)\}", src, re.S).group(1).replace("\\", " ")
    k = [np.float32(v.strip().rstrip("f")) for v in body.split(",")]
    assert len(k) == 13 and max(abs(float(v)) for v in k) <= 0.5
    x = np.concatenate([np.linspace(-12, 12, 600001), np.array([0.0, -0.0, 1e-6, -1e-6, 4.75, -4.75, 100.0, -100.0])]).astype(np.float32)
    a = np.minimum(np.abs(x), c)
    t = (a.astype(np.float64) * np.float64(np.float32(2.0) / c) - 1.0).astype(np.float32)       # one fma
    p = np.full_like(t, k[12])
    for kk in k[11::-1]:
        p = (p.astype(np.float64) * t + np.float64(kk)).astype(np.float32)                      # fma: one rounding
    m = np.abs(x) * p
    y = (x.astype(np.float64) * 0.5 + m).astype(np.float32)
    xd = x.astype(np.float64)
    want = xd * 0.5 * (1.0 + erf(xd / np.sqrt(2.0)))
    err = np.abs(y - want)
    assert err.max() < 1e-5, (err.max(), x[err.argmax()])
    assert np.abs(y[x <= -c]).max() < 2e-6 and np.abs(y[x >= c] - x[x >= c]).max() < 2e-5
    assert y[np.abs(x) < 1e-5].max() < 1e-5


def test_xcd_blocked_tile_walk_is_a_permutation_and_blocks_are_rectangles():
    """IGemmArgs::walk_bn (csrc/igemm_device.h tile_of, restated): the XCD-contiguous workgroup index -> (tile_m, tile_n) map of the
    blocked 2-D walk visits every tile exactly once, and the walk_per consecutive indices an XCD owns form one walk_tmb x walk_tnb
    rectangle - for every block shape and inner order the launcher can choose (igemm_kernel.hip walk_plan)"""
    def tile_of(wg, per, bn, tmb, tnb, n_major):
        x, i = divmod(wg, per)
        bmi, bni = divmod(x, bn)
        if n_major:
            tn, tm = divmod(i, tmb)
        else:
            tm, tn = divmod(i, tnb)
        return bmi * tmb + tm, bni * tnb + tn

    for ntm, ntn in ((16, 32), (32, 8), (16, 16), (64, 4), (8, 40), (256, 2)):
        T = ntm * ntn
        for bn in (4, 2):
            bm = 8 // bn
            if ntm % bm or ntn % bn:
                continue
            tmb, tnb, per = ntm // bm, ntn // bn, T // 8
            assert tmb * tnb == per
            for n_major in (0, 1):
                tiles = [tile_of(w, per, bn, tmb, tnb, n_major) for w in range(T)]
                assert sorted(tiles) == [(m, n) for m in range(ntm) for n in range(ntn)], (ntm, ntn, bn, n_major)
                for x in range(8):
                    blk = tiles[x * per:(x + 1) * per]
                    ms, ns = {m for m, _ in blk}, {n for _, n in blk}
                    assert len(ms) == tmb and len(ns) == tnb and max(ms) - min(ms) == tmb - 1 and max(ns) - min(ns) == tnb - 1


def test_walk_plan_takes_the_blocked_walk_where_the_round_count_says_so():
    """the launcher's choice (igemm_kernel.hip walk_plan, probed on the host - no GPU): the SDXL GEGLU (M = 4096, N = 10240, K = 1280
    on 256 x 320 tiles, one workgroup per CU, N-major by default) goes to 2 x 4 blocks of 8 x 8 tiles, N-major inside - the form
    whose count, 8 x 2 x (8 x 655 KB + 4 x 819 KB) + 42 MB of writes = 178 MB, the PMC fold measured as 178.8 MB per launch
    (profiles/r05/pmc_per_launch_class_sdxl_rows4_blocked_walk.txt); the 128 x 160 grids of the M = 4096 x N = 1280 class
    (32 x 8 tiles: block forms save 7 %) and grids that are not a multiple of 8 tiles keep their 1-D walk"""
    import ctypes as C
    from cfgpp_amd import _lib
    lib = _lib.load()
    out = (C.c_int * 5)()
    lib.cfgpp_igemm_walk_plan_probe(4096, 10240, 1280, 256, 320, 155136, 1, out)
    assert list(out) == [2, 4, 8, 8, 1], list(out)
    lib.cfgpp_igemm_walk_plan_probe(4096, 1280, 5120, 128, 160, 150000, 0, out)      # FF-out on the 16x16x32 tile
    assert out[0] == 0 and out[1] == 0, list(out)
    lib.cfgpp_igemm_walk_plan_probe(4096, 3840, 1280, 256, 256, 140000, 0, out)      # QKV: 16 x 15 tiles, no block form divides it
    assert out[0] == 0, list(out)
    lib.cfgpp_igemm_walk_plan_probe(1000, 960, 320, 256, 320, 155136, 0, out)        # 4 x 3 = 12 tiles: not a multiple of 8
    assert out[0] == 0, list(out)


def test_persistent_tile_sequence_covers_every_tile_exactly_once():
    """big4p_kernel.hip: workgroup b of G = min(T, 256) walks tiles tile_index(0), tile_index(1), ... until -1.  Mirror of the
    kernel's two forms - XCD x walks ITS contiguous eighth of the sequence G/8 tiles at a time (T % 8 == 0 and G % 8 == 0), else plain
    rounds of G over the XCD-contiguous numbering - checked for: every tile exactly once, a workgroup's rounds are consecutive
    (no gap before the -1), and in the XCD form every tile of XCD x lies in x's eighth (what keeps walk_plan's L2 model valid)."""
    def seq(T, G, b):
        xcd, idx = b & 7, b >> 3
        by_xcd = T % 8 == 0 and G % 8 == 0
        per, gx = T >> 3, G >> 3
        out, r = [], 0
        while True:
            if by_xcd:
                i = r * gx + idx
                t = xcd * per + i if i < per else -1
            else:
                q, rr = G >> 3, G & 7
                w0 = (xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q) + idx
                i = r * G + w0
                t = i if i < T else -1
            if t < 0:
                return out
            out.append(t)
            r += 1
    for T in (1, 7, 8, 27, 100, 240, 255, 256, 257, 299, 512, 640, 1280, 2560, 2563, 4096):
        G = min(T, 256)
        seen = []
        for b in range(G):
            s = seq(T, G, b)
            assert s, (T, b)                       # every launched workgroup has at least one tile
            seen += s
            if T % 8 == 0 and G % 8 == 0:
                per = T // 8
                assert all((b & 7) * per <= t < ((b & 7) + 1) * per for t in s), (T, b)
        assert sorted(seen) == list(range(T)), T
