"""Index model of igemm_epilogue_heads_staged16 (csrc/igemm_kernel.hip): one wave, every lane, LDS staging and the 16-byte
pieces, checked against the head-major layout the attention kernel reads.  python scripts/models/heads16_index_model.py"""
import itertools
def vt_pos(t): return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
def run(part_width, head_dim, heads, part0, nw0, mw0, HW, vt_linear, tok_pad, q_tok_pad, hdp):
    BLK, QKP, VTP = 1536, 48, 80
    lds = {}
    b, tok0 = mw0 // HW, mw0 % HW
    val = lambda tok, n: (tok, n)          # acc value identity: (absolute token row m, absolute column n)
    for lane in range(64):
        c16, fq = lane & 15, lane >> 4
        for j in range(5):
            ng = nw0 + j * 16; part = ng // part_width + part0; n = ng + 4 * fq
            for i in range(2):
                r = i * 16 + c16
                for k in range(4):
                    v = val(mw0 + r, n + k)      # C layout: col = lane&15 -> pixel, row = 4*(lane>>4)+k -> feature
                    if part == 2:
                        pos = r if vt_linear else vt_pos(r)
                        lds[(j * BLK + (4 * fq + k) * VTP + pos * 2)] = v
                    else:
                        lds[(j * BLK + r * QKP + (4 * fq) * 2 + 2 * k)] = v
    out = {}
    for lane in range(64):
        for j in range(5):
            ng = nw0 + j * 16; part = ng // part_width + part0
            if part == 2:
                r, c4 = lane >> 2, lane & 3
                cn = (ng + r) % part_width; head = cn // head_dim; dd = cn - head * head_dim; bh = b * heads + head
                for e in range(8):
                    out[('vt', (bh * hdp + dd) * tok_pad + tok0 + 8 * c4 + e)] = lds[j * BLK + r * VTP + c4 * 16 + 2 * e]
            else:
                r, c2 = lane >> 1, lane & 1
                cn = (ng + 8 * c2) % part_width; head = cn // head_dim; dd = cn - head * head_dim; bh = b * heads + head
                tp = q_tok_pad if part == 0 else tok_pad
                for e in range(8):
                    out[('q' if part == 0 else 'k', (bh * tp + tok0 + r) * hdp + dd + e)] = lds[j * BLK + r * QKP + c2 * 16 + 2 * e]
    # expected: every (token, column) of the slab lands where the attention kernel expects it
    n_ok = 0
    for r in range(32):
        for c in range(80):
            m, n = mw0 + r, nw0 + c
            part = n // part_width + part0; cn = n % part_width; head = cn // head_dim; dd = cn % head_dim; bh = b * heads + head
            tok = tok0 + r
            if part == 2:
                pos = tok if vt_linear else (tok & ~31) | vt_pos(tok & 31)
                key = ('vt', (bh * hdp + dd) * tok_pad + pos)
            else:
                tp = q_tok_pad if part == 0 else tok_pad
                key = ('q' if part == 0 else 'k', (bh * tp + tok) * hdp + dd)
            assert out.get(key) == (m, n), (r, c, key, out.get(key))
            n_ok += 1
    assert len(out) == 32 * 80, len(out)
    return n_ok
# SDXL QKV: C = 1280, d = 64, 20 heads, N = 3840; SD1.5 level 0: C = 320, d = 40, 8 heads (groups straddle heads); KV-only (part0 = 1)
for (pw, hd, nh, p0, N) in ((1280, 64, 20, 0, 3840), (320, 40, 8, 0, 960), (640, 80, 8, 0, 1920), (1280, 64, 20, 1, 2560), (320, 40, 8, 0, 320)):
    hdp = (hd + 31) // 32 * 32
    for nw0 in range(0, N, 80):
        for mw0 in (0, 32, 1024 + 96):
            for vl in (0, 1):
                run(pw, hd, nh, p0, nw0, mw0, 1024, vl, 1024, 1024, hdp)
print("heads16 index model: ok")
