// big4_kernel: the implicit GEMM of igemm_kernel.hip on ONE WAVE PER SIMD.
//
// Why (profiles/r04/micro_gemm_loop_variants_call32.txt): the 8-wave 256-wide tiles read one LDS fragment
// (ds_read_b128) per 32x32x16 MFMA and park two waves per SIMD at one barrier per K-tile; their K loop tops out at 0.67
// matrix-pipe utilisation in isolation.  Four waves of (MT*32) x (NT*32) - 128 x 128 on the 256 x 256 tile - need
// (MT + NT) / (MT * NT) fragment reads per MFMA (0.5) and 32 * (1/BM + 1/BN) LDS-DMA pieces per MFMA (0.25), the
// accumulators (256 .. 320 registers) live in the AGPR half of the 512-register file a lone wave owns, and the same loop
// measured 0.75 - 0.78 - provided the piece issue is slim, because at one wave per SIMD nothing hides it:
//   * a piece is `s_mov m0 ; global_load_lds_dwordx4 v_off32, s[base:base+1]`: wave-uniform 64-bit base of the K-tile (the 3x3
//     tap, the 64-channel block and the concat source are folded into it as scalar arithmetic) + the lane's own 32-bit byte
//     offset, which is constant over the K loop (recomputed only where the concat source changes / per tap for the fused
//     upsample) - no 64-bit vector add and no branch per piece;
//   * the last K-tile is peeled (no "is there a next tile" test around a piece);
//   * the pieces of tile kt+1 go out between the MFMAs of the first 3/4 of tile kt, one per gap.
// 64-deep K-tiles on a 2-stage ring (one tile of lookahead = >= 2300 matrix cycles, longer than an HBM miss), one barrier
// per K-tile, fragment reads one 16-deep step ahead.  LDS image, XOR swizzle (applied to the DMA's SOURCE chunk), activation
// row maps (AMODE 0 .. 3), two-source concat, and EVERY epilogue are igemm_kernel's (igemm_device.h): k is summed in the same
// order, so results are BIT-IDENTICAL to the other 32x32x16 tiles and the in-situ tuner may pin these configs.
//   config 24: 256 x 256 (waves 128 x 128)
//   config 25: 128 x 320 (waves 64 x 160; plain / head-major stores only)       config 26: 128 x 256 (waves 64 x 128)
// (256 x 320 on four waves would need 320 accumulator registers + 72 of fragments beside the loop's own: it spills.)
// Whole-K tiles only (no K-split); LDS-staged epilogues only; 32-bit offsets: big4_supports() is the admission test.
#include <type_traits>
#include "igemm.h"
#include "igemm_device.h"
#include "big4.h"

namespace {

template <int MT, int NT, int AMODE>
__global__ void __launch_bounds__(256)
big4_kernel(const IGemmArgs p) {
    constexpr int WTM = MT * 32, WTN = NT * 32, BM = 2 * WTM, BN = 2 * WTN;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int PAR_OFF = 2 * STAGE_BYTES;
    constexpr int QA = BM / 32, QB = BN / 32, NP = QA + QB;        // 8-row DMA pieces per WAVE and K-tile: activations, weights
    constexpr int NM = 4 * MT * NT;                                // MFMAs per wave and K-tile
    constexpr int SPAN = (NM * 3) / 4;                             // the pieces go out during the first 3/4 of them
    static_assert(NP <= SPAN, "one piece per MFMA gap at most");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tl_begin(p.tl);

    // ---- workgroup -> tile (XCD-contiguous numbering, M- or N-major walk: as igemm_kernel) ----
    int wg;
    {
        const int bid = blockIdx.x, nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    tile_of(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int HW = p.rows_per_batch;

    // ---- loader state: piece g = wid + 4 q covers rows [8 g, 8 g + 8) of the A-then-B row image of a stage ----
    const int r8 = lane >> 3, pc = lane & 7;
    // logical 16-byte chunk that belongs at physical slot pc of the lane's row: (row >> 1) & 7 with row = 8 (wid + 4 q) + r8
    // does not depend on q
    const unsigned sch16 = (unsigned)((pc ^ (((wid & 1) << 2) | (r8 >> 1))) << 4);
    int a_pix[QA];                 // pixel / token index of the lane's row (AMODE 3: packed (batch, y, x))
    unsigned a_off[QA], b_off[QB]; // byte offsets against the K-tile's scalar base
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        int m = m0 + (wid + 4 * q) * 8 + r8;
        m = m < p.M ? m : p.M - 1;
        if constexpr (AMODE == 0) a_pix[q] = m;
        else if constexpr (AMODE == 1) a_pix[q] = padded_pix(m, HW, p.W, p.H);
        else if constexpr (AMODE == 2) {
            const int b = qdiv(m, HW), r2 = m - b * HW, y = qdiv(r2, p.W), x = r2 - y * p.W;
            a_pix[q] = (b * (2 * p.H + 2) + 2 * y + 1 + p.ashift) * (2 * p.W + 2) + 2 * x + 1 + p.ashift;
        } else {
            const int b = qdiv(m, HW), r2 = m - b * HW, y = qdiv(r2, p.W), x = r2 - y * p.W;
            a_pix[q] = (b << 22) | (y << 11) | x;
        }
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        int n = n0 + (wid + 4 * q) * 8 + r8;
        n = n < p.N ? n : p.N - 1;
        b_off[q] = __umul24((unsigned)n, (unsigned)(p.K * 2)) + sch16;
    }
    // a_off for source `Cs2` = bytes per pixel of the current concat source (AMODE 3: also the tap's (dy, dx))
    auto set_a_off = [&](int Cs2, int dy, int dx) {
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            int pix;
            if constexpr (AMODE == 3) {
                const int b = a_pix[q] >> 22, y = (a_pix[q] >> 11) & 2047, x = a_pix[q] & 2047;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                pix = (b * (Hs + 2) + ((y + dy) >> 1) + 1) * (Ws + 2) + ((x + dx) >> 1) + 1;
            } else {
                pix = a_pix[q];
            }
            a_off[q] = __umul24((unsigned)pix, (unsigned)Cs2) + sch16;       // (launcher: pixels and bytes per pixel < 2^24)
        }
    };

    // ---- scalar gather state of the NEXT K-tile to request: (64-channel block g_cc of the concatenated input, tap g_tap) ----
    // Branch-free except for the (once per launch) concat-source switch.  A 1x1 / linear launch parks g_tap at the centre tap.
    const bool taps9 = p.taps == 9;
    const int PW = AMODE == 1 ? p.W + 2 : AMODE == 2 ? 2 * p.W + 2 : 0;          // padded row pitch the tap deltas use
    int g_cc = 0, g_tap = taps9 ? 0 : 4;
    int g_c0 = 0;                                  // first concatenated channel of the current source
    bool g_src1 = false;
    const char* g_src = reinterpret_cast<const char*>(p.a0);
    int g_Cs2 = p.C0 * 2;
    const char* w_base = reinterpret_cast<const char*>(p.w);   // + 128 bytes per K-tile
    const unsigned lds0 = (unsigned)(size_t)smem + (unsigned)wid * 1024u;
    set_a_off(g_Cs2, -1, -1);

    // scalar bases of the next K-tile to request (into ring stage `stage`) + advance.  The concat-source switch is taken at the
    // START of the call for the tile that needs it: every piece of the previous tile has been issued by then, so a_off may change.
    struct TileBase { const char* a; const char* b; unsigned lds; };
    auto next_tile = [&](int stage) {
        if (!g_src1 && p.C1 > 0 && g_cc >= p.C0) {
            g_src1 = true; g_src = reinterpret_cast<const char*>(p.a1); g_c0 = p.C0;
            if (p.C1 != p.C0) { g_Cs2 = p.C1 * 2; if constexpr (AMODE != 3) set_a_off(g_Cs2, 0, 0); }
        }
        TileBase t;
        const int t3 = (g_tap * 11) >> 5;                          // g_tap / 3 for 0 .. 8
        const int dy = t3 - 1, dx = g_tap - 3 * t3 - 1;
        long tap_off = 0;
        if constexpr (AMODE == 1 || AMODE == 2) tap_off = (long)((dy * PW + dx) * g_Cs2);
        if constexpr (AMODE == 3) set_a_off(g_Cs2, dy, dx);
        t.a = g_src + tap_off + (g_cc - g_c0) * 2;
        t.b = w_base;
        t.lds = lds0 + (unsigned)stage * STAGE_BYTES;
        w_base += 128;
        const int tn = g_tap + 1;
        const bool wrap = !taps9 || tn == 9;
        g_tap = !taps9 ? 4 : (wrap ? 0 : tn);
        g_cc += wrap ? 64 : 0;
        return t;
    };
    // piece g = wid + 4 q -> stage + g * 1024 (the A pieces fill exactly the BM * 128 bytes in front of the weight rows)
    auto issue_piece = [&](const TileBase& t, int q) {             // q compile-time after unrolling
        const unsigned m0v = t.lds + (unsigned)q * 4096u;
        if (q < QA) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(a_off[q]), "s"(t.a) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(b_off[q - QA]), "s"(t.b) : "memory");
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int a_rd = (wm * WTM + l31) * 128, b_rd = (BM + wn * WTN + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int t32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + t32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };

    const int nk = p.K >> 6;
    tl_stamp(p.tl, 8);
    par_stage<BN, 4>(p, smem + PAR_OFF, n0, m0, wid, lane);        // oldest loads of the kernel: the first vmcnt(0) covers them
    {
        const TileBase t0 = next_tile(0);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(t0, q);
    }
    tl_stamp(p.tl, 9);

    // one K-tile: "tile kt has landed" + barrier (which also says every wave is done reading tile kt-1, whose stage the pieces
    // of tile kt+1 overwrite), then NM MFMAs.  Placement is pinned (sched_barrier): after the FIRST MFMA of a 16-deep step the
    // next step's fragments are requested (they have MT * NT - 1 MFMAs to land); the next tile's scalar bases are computed under
    // the first step; its pieces go out one per gap, after every (SPAN / NP)-th MFMA.
    auto tile = [&](auto more_c, int stage, bool first) {
        constexpr bool MORE = decltype(more_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (first) tl_stamp(p.tl, 1);
        const char* st = smem + stage * STAGE_BYTES;
        half8_t xa[2][MT], wb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[0][i] = frag(st, a_rd, i, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) wb[0][j] = frag(st, b_rd, j, 0);
        TileBase tn = {nullptr, nullptr, 0};
        if constexpr (MORE) tn = next_tile(stage ^ 1);             // (scalar; AMODE 3: + the tap's offsets) under the first reads' latency
        int issued = 0, done = 0;                              // compile-time after unrolling
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[ks & 1][j], xa[ks & 1][i], acc[i][j], 0, 0, 0);
                    ++done;
                    if (i == 0 && j == 0 && ks + 1 < 4) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ii = 0; ii < MT; ++ii) xa[(ks + 1) & 1][ii] = frag(st, a_rd, ii, ks + 1);
#pragma unroll
                        for (int jj = 0; jj < NT; ++jj) wb[(ks + 1) & 1][jj] = frag(st, b_rd, jj, ks + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (MORE) {
                        if (issued < NP && done * NP >= (issued + 1) * SPAN) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(tn, issued); ++issued;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
        }
    };
    using T_ = std::integral_constant<bool, true>; using F_ = std::integral_constant<bool, false>;
    int stage = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        tile(T_{}, stage, kt == 0);
        if (kt == 0) tl_stamp(p.tl, 7);
        stage ^= 1;
    }
    tile(F_{}, stage, nk == 1);
    __syncthreads();                                           // every wave is done with the ring: LDS is free for the epilogue's staging
    tl_stamp(p.tl, 2);

    const int mw0 = m0 + wm * WTM, nw0 = n0 + wn * WTN;
    Par par;
    par.lds = smem + PAR_OFF; par.n0 = n0; par.b0 = HW > 0 ? qdiv(m0, HW) : 0; par.bnp = par_bnp(BN);
    // LDS-staged epilogues only: big4_supports() (the launcher's admission test) guarantees their preconditions
    if (p.epi == EPI_STORE) {
        igemm_epilogue_staged<MT, NT, true, true>(p, acc, mw0, nw0, lane, smem + wid * (32 * (WTN * 2 + 16)), par);
    } else if (p.epi == EPI_GEGLU) {
        if constexpr (NT % 2 == 0) igemm_epilogue_geglu_staged<MT, NT, true, true>(p, acc, mw0, nw0, lane, smem + wid * (32 * ((NT / 2) * 64 + 16)), par);
    } else {
        igemm_epilogue_heads_staged<MT, NT, true, true>(p, acc, mw0, nw0, lane, smem + wid * (NT * 2560), par);
    }
    tl_end(p.tl);
}

template <int MT, int NT, int AMODE>
int big4_run_amode(const IGemmArgs& a, int grid, int smem, hipStream_t stream) {
    static_assert(2 * (2 * MT * 32 + 2 * NT * 32) * 128 + par_bytes(2 * NT * 32) <= 160 * 1024, "tile does not fit the LDS");
    static int attr_smem = 0;
    auto kern = big4_kernel<MT, NT, AMODE>;
    if (smem > attr_smem) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int MT, int NT>
int big4_run_tile(const IGemmArgs& a, int grid, int smem, hipStream_t stream) {
    switch (a.amode) {
        case 0: return big4_run_amode<MT, NT, 0>(a, grid, smem, stream);
        case 1: return big4_run_amode<MT, NT, 1>(a, grid, smem, stream);
        case 2: return big4_run_amode<MT, NT, 2>(a, grid, smem, stream);
        case 3: return big4_run_amode<MT, NT, 3>(a, grid, smem, stream);
        default: cfgpp_set_error("igemm: bad amode %d", a.amode); return -2;
    }
}

}  // namespace

int big4_par_bytes(int BN, int nb) { return par_bytes(BN, nb > PAR_NB ? nb : PAR_NB); }

int big4_run(int cfg, const IGemmArgs& a, int grid, int smem, hipStream_t stream) {
    switch (cfg) {
        case 24: return big4_run_tile<4, 4>(a, grid, smem, stream);
        case 25: return big4_run_tile<2, 5>(a, grid, smem, stream);
        case 26: return big4_run_tile<2, 4>(a, grid, smem, stream);
        default: cfgpp_set_error("igemm: bad one-wave-per-SIMD config %d", cfg); return -2;
    }
}
