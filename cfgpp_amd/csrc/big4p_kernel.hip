// big4p_kernel: the 256 x 256 one-wave-per-SIMD tile of big4_kernel.hip as a PERSISTENT kernel for token-major linears
// (config 28): one workgroup per CU walks several output tiles, and the first K-tile of the NEXT output tile is requested
// (LDS-DMA) in the MFMA gaps of the current tile's last K-tile, so it lands under the epilogue.
//
// Why (profiles/r06/yardstick.txt, profiles/r05/launch_decomposition.md): on the short-K, many-tile projections - the GEGLU
// input projections (K = 320 .. 1280, 640 .. 2560 tiles of 256 x 256) - every output tile of the one-tile-per-workgroup kernels
// pays a ~4 us prologue (index math, parameter rows, the first ring stage's round trip to HBM / L2) before its first MFMA and
// the same again when the next workgroup takes the CU; with 5 .. 20 K-tiles per output tile that is 10 - 40 % of the tile.
// hipBLASLt's kernel for these shapes is persistent (Custom_Cijk_..._SK3_MT256x256x64): 1050 TF/s against 815 - 890 on
// M = 16384, N = 5120, K = 640.  Here:
//   * grid = min(tiles, 256) workgroups; XCD x walks ITS contiguous eighth of the tile sequence (the order the hardware
//     dispatcher gives the one-tile kernels, so the XCD-blocked walk of igemm_launch keeps its L2 behaviour) 32 tiles at a time;
//   * ring of two 64-deep stages as in big4_kernel; the last K-tile of output tile i runs in stage s, its MFMA gaps carry the
//     pieces of K-tile 0 of output tile i + 1 into stage s ^ 1 (offsets recomputed just before: every piece of tile i is out by
//     then), the epilogue stages through stage s (free after the barrier), and tile i + 1 starts with its K-tile 0 landed;
//   * the epilogue parameter rows (bias) are double-buffered behind the ring: tile i + 1's rows are requested before tile i's
//     epilogue reads its own.
// K is summed in the same order as in every other 32x32x16 tile (64-deep K-tiles, four 16-deep steps): results are
// BIT-IDENTICAL to them, so the in-situ tuner may pin this config.  Token-major A only (amode 0, one source, no taps);
// LDS-staged epilogues of igemm_device.h (plain store + bias + residual, GEGLU, head-major); whole-K tiles only.
#include <type_traits>
#include "igemm.h"
#include "igemm_device.h"
#include "big4.h"

namespace {

// WM x WN waves of (MT * 32) x (NT * 32); EPI compile-time (ONE epilogue body inside the tile loop).  Shipped (config 28): 2 x 4
// waves of 128 x 64, accumulators in VGPRs, two waves per SIMD.  The one-wave-per-SIMD form (2 x 2 waves of 128 x 128, 256
// accumulator AGPRs: big4_kernel's K loop) compiles from the same template (<2, 2, 4, 4>, PIN) without spills and was measured
// beside it as config 29 (call 8): never the fastest alone, never pinned in situ (its four-wave epilogue is twice as long and
// nothing overlaps it) - not instantiated.
// Register allocation inside a tile LOOP is fragile - what did NOT work, each 140 .. 690 spills: the K loop peeled into "inner
// K-tiles / last K-tile with prefetch / last K-tile without" (three copies of the MFMA body, the epilogue behind the last two);
// a literal zero fill of the accumulators (loop-invariant: hoisted as MT * NT * 16 live registers of zeros); pins, asm MFMAs with
// "+a" operands.  What works: ONE loop over (output tile, K-tile) with ONE copy of the K-tile body, an opaque zero, an opaque lane id
// in front of the epilogue.
template <int WM, int WN, int MT, int NT, int EPI>
__global__ void __launch_bounds__(64 * WM * WN)
big4p_kernel(const IGemmArgs p, const int par_buf_bytes) {
    constexpr int NWV = WM * WN;
    constexpr bool PIN = NWV == 4;                                 // one wave per SIMD: accumulators in the AGPR file (igemm_device.h: acc_row_pin)
    constexpr int WTM = MT * 32, WTN = NT * 32, BM = WM * WTM, BN = WN * WTN;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int PAR_OFF = 2 * STAGE_BYTES;
    constexpr int QA = BM / (8 * NWV), QB = BN / (8 * NWV), NP = QA + QB;      // 8-row DMA pieces per WAVE and K-tile: activations, weights
    constexpr int NM = 4 * MT * NT;                                // MFMAs per wave and K-tile
    constexpr int SPAN = (NM * 3) / 4;                             // the pieces go out during the first 3/4 of them
    static_assert(NP <= SPAN, "one piece per MFMA gap at most");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid - wm * WN;
    const int HW = p.rows_per_batch;

    // ---- the workgroup's tile sequence ----
    const int T = p.n_main, G = gridDim.x;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const bool by_xcd = (T & 7) == 0 && (G & 7) == 0;
    const int per = T >> 3, gx = G >> 3;                           // tiles / resident workgroups per XCD
    int round = 0;
    auto tile_index = [&](int r) {                                 // -1: no such tile
        if (by_xcd) { const int i = r * gx + idx; return i < per ? xcd * per + i : -1; }
        const int q = G >> 3, rr = G & 7;
        const int w0 = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
        const int i = r * G + w0;
        return i < T ? i : -1;
    };

    // ---- loader state of the output tile whose K-tiles are being requested ----
    const int r8 = lane >> 3, pc = lane & 7;
    const unsigned sch16 = (unsigned)((pc ^ (((wid & 1) << 2) | (r8 >> 1))) << 4);
    unsigned a_off[QA], b_off[QB];                                 // byte offsets against the K-tile's scalar bases
    int m0 = 0, n0 = 0;
    const char* a_base = nullptr; const char* w_base = nullptr;    // + 128 bytes per K-tile
    const unsigned row_bytes = (unsigned)(p.K * 2);
    auto setup_tile = [&](int wg) {
        int tile_m, tile_n;
        tile_of(p, wg, tile_m, tile_n);
        m0 = tile_m * BM; n0 = tile_n * BN;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            int m = m0 + (wid + NWV * q) * 8 + r8;
            m = m < p.M ? m : p.M - 1;
            a_off[q] = __umul24((unsigned)m, row_bytes) + sch16;   // (launcher: rows and bytes per row < 2^24)
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            int n = n0 + (wid + NWV * q) * 8 + r8;
            n = n < p.N ? n : p.N - 1;
            b_off[q] = __umul24((unsigned)n, row_bytes) + sch16;
        }
        a_base = reinterpret_cast<const char*>(p.a0);
        w_base = reinterpret_cast<const char*>(p.w);
    };
    const unsigned lds0 = (unsigned)(size_t)smem + (unsigned)wid * 1024u;
    struct TileBase { const char* a; const char* b; unsigned lds; };
    auto next_tile = [&](int stage) {
        TileBase t;
        t.a = a_base; t.b = w_base; t.lds = lds0 + (unsigned)stage * STAGE_BYTES;
        a_base += 128; w_base += 128;
        return t;
    };
    auto issue_piece = [&](const TileBase& t, int q) {             // q compile-time after unrolling
        const unsigned m0v = t.lds + (unsigned)q * (1024u * NWV);
        if (q < QA) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(a_off[q]), "s"(t.a) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(b_off[q - QA]), "s"(t.b) : "memory");
    };

    f32x16 acc[MT][NT];
    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int a_rd = (wm * WTM + l31) * 128, b_rd = (BM + wn * WTN + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int t32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + t32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };
    const int nk = p.K >> 6;

    // one K-tile (big4_kernel's): "the tile has landed" + barrier, then NM MFMAs with the NEXT K-tile's pieces in their gaps
    auto tile = [&](auto more_c, int stage) {
        constexpr bool MORE = decltype(more_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + stage * STAGE_BYTES;
        half8_t xa[2][MT], wb[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[0][i] = frag(st, a_rd, i, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) wb[0][j] = frag(st, b_rd, j, 0);
        TileBase tn = {nullptr, nullptr, 0};
        if constexpr (MORE) tn = next_tile(stage ^ 1);
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

    // ---- first output tile: parameter rows, then K-tile 0 into stage 0 ----
    int wg = tile_index(0);
    if (wg < 0) return;                                            // (grid <= tiles: cannot happen; wave-uniform anyway)
    int pbuf = 0;
    setup_tile(wg);
    par_stage<BN, NWV>(p, smem + PAR_OFF, n0, m0, wid, lane);
    {
        const TileBase t0 = next_tile(0);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(t0, q);
    }
    // ONE loop over (output tile, K-tile) with ONE copy of the K-tile body: K-tile kt always carries the pieces of "the next K-tile" in
    // its MFMA gaps - of the same output tile, or (kt == nk - 1) K-tile 0 of the NEXT output tile, whose offsets are computed just
    // before (every piece of the current tile is out by then).  After the last output tile the "next" is this tile's own K-tile 0
    // again: a harmless 64 KB re-read into the idle stage instead of a branch around every piece; drained before the kernel ends.
    int stage = 0, kt = 0;
    int cm0 = m0, cn0 = n0, cbuf = 0;                              // the tile being accumulated (m0 / n0 move on one K-tile early)
    bool last_tile = false;
    float zero = 0.f;
    asm volatile("" : "+v"(zero));                                 // (opaque: a literal fill would be hoisted as MT * NT * 16 live registers of zeros)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = zero;
    for (;;) {
        const bool tile_end = kt == nk - 1;                        // wave-uniform
        if (tile_end) {
            const int wg_next = tile_index(++round);
            last_tile = wg_next < 0;
            if (!last_tile) {
                setup_tile(wg_next);
                pbuf ^= 1;
                par_stage<BN, NWV>(p, smem + PAR_OFF + pbuf * par_buf_bytes, n0, m0, wid, lane);   // (older than the pieces that follow)
            } else {
                a_base = reinterpret_cast<const char*>(p.a0); w_base = reinterpret_cast<const char*>(p.w);
            }
        }
        tile(T_{}, stage);
        if (tile_end) {
            __syncthreads();                                       // every wave is done reading `stage`: it is the epilogue's staging area now
            int lane_e = threadIdx.x & 63;
            asm volatile("" : "+v"(lane_e));                       // (opaque per tile: lane-derived epilogue addresses are not hoisted out of the loop)
            const int mw0 = cm0 + wm * WTM, nw0 = cn0 + wn * WTN;
            char* const stg = smem + stage * STAGE_BYTES;
            Par par;
            par.lds = smem + PAR_OFF + cbuf * par_buf_bytes; par.n0 = cn0; par.b0 = HW > 0 ? qdiv(cm0, HW) : 0; par.bnp = par_bnp(BN);
            if constexpr (EPI == EPI_STORE) {
                igemm_epilogue_staged<MT, NT, true, PIN>(p, acc, mw0, nw0, lane_e, stg + wid * (32 * (WTN * 2 + 16)), par);
            } else if constexpr (EPI == EPI_GEGLU) {
                igemm_epilogue_geglu_staged<MT, NT, true, PIN>(p, acc, mw0, nw0, lane_e, stg + wid * (32 * ((NT / 2) * 64 + 16)), par);
            } else {
                igemm_epilogue_heads_staged<MT, NT, true, PIN>(p, acc, mw0, nw0, lane_e, stg + wid * (NT * 2560), par);
            }
            if (last_tile) break;
            cm0 = m0; cn0 = n0; cbuf = pbuf; kt = -1;
            asm volatile("" : "+v"(zero));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = zero;
        }
        ++kt;
        stage ^= 1;                                                // (after an epilogue: the next output tile's K-tile 0 is arriving there; its
                                                                   //  tile() waits for it and its barrier retires every wave's staging reads)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the dummy prefetch: no LDS-DMA may land after this workgroup's LDS is handed on
}

}  // namespace

// LDS: the ring + TWO parameter-row buffers
int big4p_smem(int par_bytes_one) { return 2 * (256 + 256) * 128 + 2 * par_bytes_one; }

namespace {
template <int WM, int WN, int MT, int NT, int EPI>
int big4p_run_epi(const IGemmArgs& a, int grid, int smem, int par_bytes_one, hipStream_t stream) {
    static int attr_smem = 0;
    auto kern = big4p_kernel<WM, WN, MT, NT, EPI>;
    if (smem > attr_smem) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), smem, stream, a, par_bytes_one);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int WM, int WN, int MT, int NT>
int big4p_run_form(const IGemmArgs& a, int grid, int smem, int par_bytes_one, hipStream_t stream) {
    switch (a.epi) {
        case EPI_STORE: return big4p_run_epi<WM, WN, MT, NT, EPI_STORE>(a, grid, smem, par_bytes_one, stream);
        case EPI_GEGLU: return big4p_run_epi<WM, WN, MT, NT, EPI_GEGLU>(a, grid, smem, par_bytes_one, stream);
        default: return big4p_run_epi<WM, WN, MT, NT, EPI_HEADS>(a, grid, smem, par_bytes_one, stream);
    }
}
}  // namespace

int big4p_run(const IGemmArgs& a, int grid, int smem, int par_bytes_one, hipStream_t stream) {
    return big4p_run_form<2, 4, 4, 2>(a, grid, smem, par_bytes_one, stream);
}
