"""The LDS layout claims of DESIGN.md section 3.1, checked on the CPU against the bank model of the MI355X guide: a wave64
``ds_read_b128`` is serviced in four fixed 16-lane groups, the bank of byte address a is (a / 4) mod 64, a group is
conflict-free when its 16 sixteen-byte reads hit 16 distinct bank quads.

Tiles are [rows][64 halfs] (128-byte rows) with the 16-byte-chunk swizzle ``physical = logical ^ ((row >> 1) & 7)``
(csrc/igemm_kernel.hip; the LDS-DMA loader applies it to the SOURCE chunk because the DMA stores lane-linear)."""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _quads(addr_of_lane):
    """per lane group: the set of (address / 16) mod 16 bank quads its lanes touch"""
    return [[(addr_of_lane(l) // 16) % 16 for l in g] for g in B128_GROUPS]


def _conflict_free(addr_of_lane):
    return all(len(set(q)) == 16 for q in _quads(addr_of_lane))


def test_lane_groups_partition_the_wave():
    assert sorted(itertools.chain.from_iterable(B128_GROUPS)) == list(range(64)) and all(len(g) == 16 for g in B128_GROUPS)


def test_32x32x16_fragment_reads_are_conflict_free():
    """igemm_kernel / attn64_kernel: row = base + (lane & 31), logical chunk = 2 * ks + (lane >> 5)"""
    for base in range(0, 320, 32):                      # every 32-row sub-tile of the largest tile
        for ks in range(4):
            def addr(l, base=base, ks=ks):
                row = base + (l & 31)
                return row * 128 + (((2 * ks + (l >> 5)) ^ ((row >> 1) & 7)) << 4)
            assert _conflict_free(addr), (base, ks)


def test_16x16x32_fragment_reads_are_conflict_free():
    """igemm16_kernel: row = base + (lane & 15), logical chunk = 4 * s + (lane >> 4); bases are multiples of 16 (wave
    slabs start at multiples of 32 rows of activations / 80 rows of weights, sub-tiles every 16 rows)"""
    for base in range(0, 160, 16):
        for s in range(2):
            def addr(l, base=base, s=s):
                row = base + (l & 15)
                return row * 128 + (((4 * s + (l >> 4)) ^ ((row >> 1) & 7)) << 4)
            assert _conflict_free(addr), (base, s)


def test_unswizzled_rows_would_conflict():
    """the control: without the XOR the same reads serialise (what the swizzle is for)"""
    def addr(l):
        return (l & 31) * 128 + ((l >> 5) << 4)
    worst = max(max(q.count(v) for v in set(q)) for q in _quads(addr))
    assert worst >= 8


def test_dma_source_swizzle_matches_fragment_reads():
    """loader: lane (lrow = tid >> 3, lchunk = tid & 7) stores LINEARLY to row r, physical slot lchunk, and loads the source
    chunk lchunk ^ ((r >> 1) & 7); a fragment read of logical chunk c at row r looks at physical slot c ^ ((r >> 1) & 7): it must
    find logical chunk c there."""
    for r in range(0, 288):
        stored = {phys: phys ^ ((r >> 1) & 7) for phys in range(8)}       # physical slot -> logical chunk the DMA put there
        for c in range(8):
            assert stored[c ^ ((r >> 1) & 7)] == c


def test_vt_key_permutation():
    """cfgpp_vt_pos swaps bits 2 and 3 of the key index inside every 32-key block: an involution, and the 8 keys a lane
    feeds to one PV MFMA (16 * tt + 8 * b + 4 * hi + r, b = 0..1, r = 0..3) become 8 CONSECUTIVE positions = one 16-byte read"""
    def pos(t):
        return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    assert sorted(pos(t) for t in range(32)) == list(range(32)) and all(pos(pos(t)) == t for t in range(32))
    for tt in range(2):
        for hi in range(2):
            keys = [16 * tt + 8 * b + 4 * hi + r for b in range(2) for r in range(4)]
            p = sorted(pos(k) for k in keys)
            assert p == list(range(p[0], p[0] + 8)) and p[0] % 8 == 0, (tt, hi, p)


# ---- round 4: 64-byte rows (32-deep K-tiles of tile32_kernel), the conv_out halo tile ----
def test_64_byte_rows_fragment_reads_and_dma_swizzle():
    """tile32_kernel: rows of 64 bytes (4 chunks), physical = logical ^ ((row >> 2) & 3); a
    fragment read is row = base + (lane & 31), logical chunk 2 * ks + (lane >> 5), ks = 0, 1; the DMA piece is 16 rows, lane ->
    (row = lane >> 2, physical slot = lane & 3) loading source chunk slot ^ ((row >> 2) & 3)"""
    for base in range(0, 384, 32):
        for ks in range(2):
            def addr(l, base=base, ks=ks):
                row = base + (l & 31)
                return row * 64 + (((2 * ks + (l >> 5)) ^ ((row >> 2) & 3)) << 4)
            assert _conflict_free(addr), (base, ks)
    for piece in range(24):
        for lane in range(64):
            r = piece * 16 + (lane >> 2)
            src_chunk = (lane & 3) ^ (((lane >> 2) >> 2) & 3)          # what the kernel computes from the lane alone
            assert src_chunk == (lane & 3) ^ ((r >> 2) & 3)           # = the swizzle of the tile row the piece lands on
            assert (src_chunk ^ ((r >> 2) & 3)) == (lane & 3)         # a read of logical `src_chunk` at row r finds slot lane & 3

    def unswizzled(l):
        return (l & 31) * 64 + ((l >> 5) << 4)
    assert max(max(q.count(v) for v in set(q)) for q in _quads(unswizzled)) >= 4


def test_conv_out_halo_tile_reads_are_conflict_free():
    """conv_out_tile_kernel: thread (ty, tx) = pixel of an 8 x 32 tile reads 16-byte chunk ch of halo pixel (ty + dy, tx + dx) from
    a [10 x 34 pixels][144 bytes] tile; a wave = two tile rows"""
    for dy in range(3):
        for dx in range(3):
            for ch in range(8):
                def addr(l, dy=dy, dx=dx, ch=ch):
                    ty, tx = l >> 5, l & 31
                    return ((ty + dy) * 34 + tx + dx) * 144 + ch * 16
                assert _conflict_free(addr), (dy, dx, ch)
