// Implicit-GEMM MFMA kernel for gfx950: conv3x3 (stride 1 / stride 2 / fused
// nearest-2x upsample), conv1x1 and linear layers of the SD / SDXL UNet, all as
//      D[n][m] = sum_k W[n][k] * X[m][k]           (fp16 in, fp32 accumulate)
// with the activation rows X gathered on the fly from halo-padded NHWC (3x3
// taps are plain pointer deltas, no bounds checks) or token-major tensors, and
// from up to two channel-concatenated sources (skip-connection concat is never
// materialised).
//
// CDNA4 mapping
//   * v_mfma_f32_32x32x16_f16, "swapped" operands: MFMA-A = weights (rows = output
//     feature n), MFMA-B = activations (cols = pixel/token m).  The accumulator then
//     holds, per lane, 4 CONSECUTIVE output features of ONE pixel -> 8-byte fp16
//     stores into row-major [m][n] outputs and cheap per-feature bias loads.
//   * 64x64 (or 32x32) tile per wave, WM x WN waves per workgroup, BK = 64.
//   * LDS tiles [rows][64 halfs] (128-B rows) with a 16-B-chunk XOR swizzle
//     chunk ^= (row>>1)&7: ds_write_b128 (8-lane groups) and ds_read_b128 (16-lane
//     groups of distinct rows) are both bank-conflict free.
//   * register-staged double buffering: global loads of k-tile t+1 are issued before
//     the MFMAs of tile t and written to the other LDS stage afterwards; one barrier
//     per k-tile.
//   * XCD-aware workgroup remap: consecutive tiles (sharing activation rows) stay on
//     one XCD's L2.
// Epilogues: bias, per-batch time-embedding add, residual add, GEGLU, and the
// head-major Q / K / V^T scatter the attention kernel consumes.
#include <cstring>
#include <map>
#include <tuple>
#include <type_traits>
#include "igemm.h"

#include "igemm_device.h"
#include "big4.h"

namespace {

// GLDS = true: tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write pass).  The DMA writes lane-linear (wave base + lane*16 B), so the XOR swizzle is
// applied to the per-lane SOURCE chunk instead (same 128-B segment: coalescing unchanged).
// AMODE (activation row map) is a template parameter: the k-loop must not branch on it.
template <int WM, int WN, int WTM, int WTN, bool GLDS, int AMODE, int NST = 2>
__global__ void __launch_bounds__(64 * WM * WN)
igemm_kernel(const IGemmArgs p) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int BM = WM * WTM, BN = WN * WTN;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int RSTEP = NTHR / 8;            // rows covered per loader pass
    constexpr int A_CH = BM / RSTEP, B_CH = BN / RSTEP;
    static_assert(BM % RSTEP == 0 && BN % RSTEP == 0, "tile/loader mismatch");
    constexpr int STAGE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    tl_begin(p.tl);
    constexpr int PAR_OFF = NST * (WM * WTM + WN * WTN) * 128;       // epilogue parameters behind the ring (LDS-DMA kernels only)

    // ---- workgroup -> (tile, k-range).  Blocks [0, n_main) own one whole tile each (XCD-aware order:
    // block b runs on XCD b % 8, consecutive tiles share activation rows).  Blocks >= n_main are the
    // K-split TAIL: tile n_main + b/ksplit, k-slice b % ksplit; they write fp32 partials that
    // igemm_reduce_kernel finishes.  (Tile quantisation, not the inner loop, is what the profile showed
    // to cost most: e.g. 640 tiles on 512 resident slots, or 80 tiles on 256 CUs.)
    const int ntn = (p.N + BN - 1) / BN;
    const int bid = blockIdx.x;
    int wg, ksl = 0;
    const bool is_tail = bid >= p.n_main;
    if (!is_tail) {
        const int nwg = p.n_main;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    } else {
        // same XCD-contiguous numbering over the (tile, k-slice) pairs: the k-slices of neighbouring tiles (which share
        // activation rows / weight slabs slice by slice) meet in one XCD's L2
        const int b1 = bid - p.n_main, nb = (int)gridDim.x - p.n_main;
        const int q = nb >> 3, r = nb & 7;
        const int xcd = b1 & 7, idx = b1 >> 3;
        const int b2 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        wg = p.n_main + b2 / p.ksplit;
        ksl = b2 - (b2 / p.ksplit) * p.ksplit;
    }
    // Tile walk.  M-major (consecutive tiles share activation rows; an XCD's L2 streams every weight slab once): right
    // when the activations are the larger operand.  N-major (consecutive tiles share a weight slab, which then stays in
    // the XCD's L2 while the - smaller - activation matrix is what every XCD streams): right for the weight-heavy
    // launches (GEGLU at M = 4096: 26 MB of weights vs 10 MB of activations; PMC showed 56 % L2 misses M-major).
    // The launcher picks by operand bytes (IGemmArgs::n_major); the result does not depend on it.
    // (one division by a launcher-provided divisor: a two-sided branch here cost the 256 x 320 kernel 30 VGPRs -> spills)
    int tile_m, tile_n;
    tile_of(p, wg, tile_m, tile_n);                        // (K-split tails never take the blocked walk: walk_bn = 0)
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;

    // ---- loader state ----
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int HW = p.rows_per_batch;
    const int Cin = p.C0 + p.C1;
    int a_pix[A_CH];
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
        int m = m0 + lrow + j * RSTEP;
        m = m < p.M ? m : p.M - 1;
        if constexpr (AMODE == 0) a_pix[j] = m;
        else if constexpr (AMODE == 1) a_pix[j] = padded_pix(m, HW, p.W, p.H);
        else if constexpr (AMODE == 2) {
            const int b = qdiv(m, HW), q = m - b * HW, y = qdiv(q, p.W), x = q - y * p.W;
            a_pix[j] = (b * (2 * p.H + 2) + 2 * y + 1 + p.ashift) * (2 * p.W + 2) + 2 * x + 1 + p.ashift;
        } else {
            const int b = qdiv(m, HW), q = m - b * HW, y = qdiv(q, p.W), x = q - y * p.W;
            a_pix[j] = (b << 22) | (y << 11) | x;      // unpacked per tap
        }
    }
    // source chunk: register staging loads logical chunk `lchunk` and swizzles the LDS store address;
    // the DMA path stores linearly, so it loads the chunk that BELONGS at physical slot `lchunk`.
    const int schunk = GLDS ? (lchunk ^ ((lrow >> 1) & 7)) : lchunk;
    const half_t* b_ptr[B_CH];
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
        int n = n0 + lrow + j * RSTEP;
        n = n < p.N ? n : p.N - 1;
        b_ptr[j] = p.w + (long)n * p.K + schunk * 8;
    }
    // LDS store offsets (swizzled), identical for both operands
    int st_off[(A_CH > B_CH ? A_CH : B_CH)];
#pragma unroll
    for (int j = 0; j < (A_CH > B_CH ? A_CH : B_CH); ++j) {
        const int r = lrow + j * RSTEP;
        st_off[j] = r * 128 + ((lchunk ^ ((r >> 1) & 7)) << 4);
    }

    half8_t ra[A_CH], rb[B_CH];
    const int KT_all = p.K >> 6;
    const int kt_begin = is_tail ? (int)(((long)ksl * KT_all) / p.ksplit) : 0;
    const int kt_end = is_tail ? (int)(((long)(ksl + 1) * KT_all) / p.ksplit) : KT_all;

    auto load_tile = [&](int kt) {
        int tap = 0, cc = kt << 6;
        if (p.taps == 9) { const int cb = kt / 9; tap = kt - cb * 9; cc = cb << 6; }
        const bool s0 = cc < p.C0;
        const half_t* src = s0 ? p.a0 : p.a1;
        const int cs = s0 ? cc : cc - p.C0, Cs = s0 ? p.C0 : p.C1;
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        int dpix = 0;
        if constexpr (AMODE == 1) dpix = dy * (p.W + 2) + dx;
        else if constexpr (AMODE == 2) dpix = dy * (2 * p.W + 2) + dx;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            int pix;
            if constexpr (AMODE == 3) {
                const int b = a_pix[j] >> 22, y = (a_pix[j] >> 11) & 2047, x = a_pix[j] & 2047;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                pix = (b * (Hs + 2) + ((y + dy) >> 1) + 1) * (Ws + 2) + ((x + dx) >> 1) + 1;
            } else {
                pix = a_pix[j] + dpix;
            }
            ra[j] = *reinterpret_cast<const half8_t*>(src + (long)pix * Cs + cs + lchunk * 8);
        }
#pragma unroll
        for (int j = 0; j < B_CH; ++j) rb[j] = *reinterpret_cast<const half8_t*>(b_ptr[j] + ((long)kt << 6));
    };
    // DMA variant: same addresses, destination = wave-uniform LDS base (+ lane*16 added by hardware)
    const int wave_row0 = __builtin_amdgcn_readfirstlane(wid) * 8;
    auto dma_tile = [&](int kt, int stage) {
        int tap = 0, cc = kt << 6;
        if (p.taps == 9) { const int cb = kt / 9; tap = kt - cb * 9; cc = cb << 6; }
        const bool s0 = cc < p.C0;
        const half_t* src = s0 ? p.a0 : p.a1;
        const int cs = s0 ? cc : cc - p.C0, Cs = s0 ? p.C0 : p.C1;
        int dy = 0, dx = 0;
        if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        int dpix = 0;
        if constexpr (AMODE == 1) dpix = dy * (p.W + 2) + dx;
        else if constexpr (AMODE == 2) dpix = dy * (2 * p.W + 2) + dx;
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * 128;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) {
            int pix;
            if constexpr (AMODE == 3) {
                const int b = a_pix[j] >> 22, y = (a_pix[j] >> 11) & 2047, x = a_pix[j] & 2047;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                pix = (b * (Hs + 2) + ((y + dy) >> 1) + 1) * (Ws + 2) + ((x + dx) >> 1) + 1;
            } else {
                pix = a_pix[j] + dpix;
            }
            const half_t* g = src + (long)pix * Cs + cs + schunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(As + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_CH; ++j) {
            const half_t* g = b_ptr[j] + ((long)kt << 6);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(Bs + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
        }
    };
    auto store_tile = [&](int stage) {
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * 128;
#pragma unroll
        for (int j = 0; j < A_CH; ++j) *reinterpret_cast<half8_t*>(As + st_off[j]) = ra[j];
#pragma unroll
        for (int j = 0; j < B_CH; ++j) *reinterpret_cast<half8_t*>(Bs + st_off[j]) = rb[j];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    // fragment read offsets: row = base + (lane&31), logical chunk = ks*2 + (lane>>5)
    const int frow = lane & 31, fhi = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    const int a_rd = (wm * WTM + frow) * 128;
    const int b_rd = (wn * WTN + frow) * 128;

    if constexpr (GLDS) {
        // ---- DMA path: the next tile's global->LDS pieces are issued BETWEEN the MFMAs of this tile ----
        // Issued back to back at the top of the tile, the (A_CH + B_CH) LDS-DMA instructions of the two waves
        // that share a SIMD cost ~100-185 cycles of issue each while neither wave feeds the matrix pipe (ISA +
        // guide "LDS-DMA piece issue cost").  Spread one by one over the first half of the tile's MFMAs they
        // cost ~60 cycles each and the partner wave's MFMAs cover them; the second half of the tile is the
        // landing time before the end-of-tile vmcnt(0) + barrier.
        constexpr int NP = A_CH + B_CH;               // DMA pieces per wave per K-tile
        constexpr int NM = 4 * MT * NT;               // MFMAs per wave per K-tile
        constexpr bool ILV = (WM * WN >= 8);          // 4-wave tiles overlap through co-resident blocks instead
        constexpr int SPAN = ILV ? (NM * 5) / 8 : 0;  // pieces go out during the first 5/8 of the MFMAs
        // NST-stage LDS ring: tile kt is computed from stage (kt - kt_begin) % NST while tiles kt+1 .. kt+NST-1
        // are in flight / landed; a tile has NST-1 tile times to arrive (memory latency under load is of the
        // order of one tile time, so NST = 2 stalls at the end-of-tile wait).
        const int nk = kt_end - kt_begin;
        tl_stamp(p.tl, 8);
        // the parameter segments are the oldest loads of the kernel: covered by every counted wait.  (Requesting them AFTER the
        // prologue's tiles instead - conservative counted waits, published by the loop's draining waits - measured the same
        // in situ, profiles/r03/ab/par_late_ln_fusion_call6.txt, and the second copy of par_stage cost the 256 x 320 kernel 100 spills.)
        if (!is_tail) par_stage<BN, WM * WN>(p, smem + PAR_OFF, n0, m0, __builtin_amdgcn_readfirstlane(wid), lane);
#pragma unroll
        for (int s_ = 0; s_ < NST - 1; ++s_) if (s_ < nk) dma_tile(kt_begin + s_, s_);
        tl_stamp(p.tl, 9);
        if (NST > 2 && nk >= NST - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST > 2 ? (NST - 2) * NP : 0)) : "memory");   // NST-1 tiles issued: the oldest has landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tl_stamp(p.tl, 1);
        // one K-tile: MFMAs on stage `cur`; DMA (if any) of tile `ktn` into stage `nxt`; wait + barrier
        auto tile_body = [&](int kt, int cur, int ktn, int nxt, auto with_dma) {
            constexpr bool DMA = decltype(with_dma)::value;
            const char* As = smem + cur * STAGE_BYTES;
            const char* Bs = As + BM * 128;
            // per-tile scalars of the NEXT tile's gather
            const half_t* src = p.a0; int cs = 0, Cs = p.C0, dpix = 0, dy = 0, dx = 0;
            char* Asn = smem + nxt * STAGE_BYTES;
            char* Bsn = Asn + BM * 128;
            if constexpr (DMA) {
                int tap = 0, cc = ktn << 6;
                if (p.taps == 9) { const int cb = ktn / 9; tap = ktn - cb * 9; cc = cb << 6; }
                const bool s0 = cc < p.C0;
                src = s0 ? p.a0 : p.a1;
                cs = s0 ? cc : cc - p.C0; Cs = s0 ? p.C0 : p.C1;
                if (p.taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
                if constexpr (AMODE == 1) dpix = dy * (p.W + 2) + dx;
                else if constexpr (AMODE == 2) dpix = dy * (2 * p.W + 2) + dx;
            }
            auto piece = [&](int q) {
                if (q < A_CH) {
                    const int j = q;
                    int pix;
                    if constexpr (AMODE == 3) {
                        const int b = a_pix[j] >> 22, y = (a_pix[j] >> 11) & 2047, x = a_pix[j] & 2047;
                        const int Hs = p.H >> 1, Ws = p.W >> 1;
                        pix = (b * (Hs + 2) + ((y + dy) >> 1) + 1) * (Ws + 2) + ((x + dx) >> 1) + 1;
                    } else {
                        pix = a_pix[j] + dpix;
                    }
                    const half_t* g = src + (long)pix * Cs + cs + schunk * 8;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(Asn + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
                } else {
                    const int j = q - A_CH;
                    const half_t* g = b_ptr[j] + ((long)ktn << 6);
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (__attribute__((address_space(3))) void*)(Bsn + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
                }
            };
            int issued = 0, done = 0;                 // compile-time after unrolling
            if constexpr (DMA && !ILV) {
#pragma unroll
                for (int q = 0; q < NP; ++q) piece(q);
                issued = NP;
            }
            auto after_mfma = [&]() {
                ++done;
                if constexpr (DMA && ILV) {
                    if (issued < NP && done * NP >= (issued + 1) * SPAN) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(issued); ++issued;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if constexpr (MT * NT <= 8) {
                // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
                half8_t xa[2][MT], wb[2][NT];
                {
                    const int coff = ((0 | fhi) ^ fsw) << 4;
#pragma unroll
                    for (int i = 0; i < MT; ++i) xa[0][i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                    for (int j = 0; j < NT; ++j) wb[0][j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) {
                        const int coff = ((((ks + 1) << 1) | fhi) ^ fsw) << 4;
#pragma unroll
                        for (int i = 0; i < MT; ++i) xa[(ks + 1) & 1][i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                        for (int j = 0; j < NT; ++j) wb[(ks + 1) & 1][j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[ks & 1][j], xa[ks & 1][i], acc[i][j], 0, 0, 0);
                            after_mfma();
                        }
                    }
                }
            } else {
                // 10 accumulator tiles per wave (256x320): no room for a second fragment set; 10 MFMAs per k-step
                // (320 cycles) give the partner wave on the SIMD time to cover the LDS latency instead
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int coff = (((ks << 1) | fhi) ^ fsw) << 4;
                    half8_t xa[MT], wb[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) xa[i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                    for (int j = 0; j < NT; ++j) wb[j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                            after_mfma();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // keep one fragment set live: no hoisting of the next k-step's reads
                }
            }
            if constexpr (DMA) {
#pragma unroll
                for (int q = 0; q < NP; ++q) if (q >= issued) piece(q);      // (none when SPAN <= NM)
            }
            if constexpr (NST == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else {
                // the next tile has landed once at most the younger tiles' pieces are outstanding.  More than one tile
                // in flight across the barrier: raw s_barrier (__syncthreads() would drain the DMA queue)
                if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        int kt = kt_begin;
        for (; kt + NST - 1 < kt_end; ++kt) {
            tile_body(kt, (kt - kt_begin) % NST, kt + NST - 1, (kt - kt_begin + NST - 1) % NST, std::true_type{});
            if (kt == kt_begin) tl_stamp(p.tl, 7);
        }
        for (; kt < kt_end; ++kt) tile_body(kt, (kt - kt_begin) % NST, 0, 0, std::false_type{});
        tl_stamp(p.tl, 2);
    } else {
        load_tile(kt_begin);
        store_tile(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) {
            load_tile(kt + 1);
        }
        const char* As = smem + cur * STAGE_BYTES;
        const char* Bs = As + BM * 128;
        if constexpr (MT * NT <= 8) {
            // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
            half8_t xa[2][MT], wb[2][NT];
            {
                const int coff = ((0 | fhi) ^ fsw) << 4;
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[0][i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                for (int j = 0; j < NT; ++j) wb[0][j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) {
                    const int coff = ((((ks + 1) << 1) | fhi) ^ fsw) << 4;
#pragma unroll
                    for (int i = 0; i < MT; ++i) xa[(ks + 1) & 1][i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                    for (int j = 0; j < NT; ++j) wb[(ks + 1) & 1][j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[ks & 1][j], xa[ks & 1][i], acc[i][j], 0, 0, 0);
            }
        } else {
            // 10 accumulator tiles per wave (256x320): no room for a second fragment set; 10 MFMAs per k-step
            // (320 cycles) give the partner wave on the SIMD time to cover the LDS latency instead
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int coff = (((ks << 1) | fhi) ^ fsw) << 4;
                half8_t xa[MT], wb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 32 * 128 + coff);
#pragma unroll
                for (int j = 0; j < NT; ++j) wb[j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 32 * 128 + coff);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // keep one fragment set live: no hoisting of the next k-step's reads
            }
        }
        if (kt + 1 < kt_end) store_tile(cur ^ 1);
        __syncthreads();
    }

    }

    const int mw0 = m0 + wm * WTM, nw0 = n0 + wn * WTN;
    if (is_tail) {
        // fp32 partial, register order, coalesced: ws[((block * REGS + r) * NTHR) + tid]
        float* ws = p.ws + ((long)(wg - p.n_main) * p.ksplit + ksl) * (MT * NT * 16) * NTHR + tid;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ws[(long)((i * NT + j) * 16 + r) * NTHR] = acc[i][j][r];
        tl_end(p.tl);
        return;
    }
    Par par;
    par.lds = GLDS ? smem + PAR_OFF : nullptr; par.n0 = n0; par.b0 = HW > 0 ? qdiv(m0, HW) : 0; par.bnp = par_bnp(BN);
    constexpr bool STAGED_FITS = WM * WN * 32 * (WTN * 2 + 16) <= NST * STAGE_BYTES;      // (256 x 320 with 32 x 320 waves: 164 KB, no)
    if (STAGED_FITS && p.epi == EPI_STORE && (p.N & 7) == 0 && p.staged_epi) {
        // the k-loop ended with a barrier: every wave is done with the tile stages, LDS is free
        igemm_epilogue_staged<MT, NT, GLDS>(p, acc, mw0, nw0, lane, smem + wid * (32 * (WTN * 2 + 16)), par);
        tl_end(p.tl);
        return;
    }
    if constexpr (NT % 2 == 0) {
        if (p.epi == EPI_GEGLU && (p.N & 127) == 0 && p.staged_epi && p.omode == 0) {
            igemm_epilogue_geglu_staged<MT, NT, GLDS>(p, acc, mw0, nw0, lane, smem + wid * (32 * ((NT / 2) * 64 + 16)), par);
            tl_end(p.tl);
            return;
        }
    }
    // (not instantiated for the 10-accumulator-tile waves: it pushed the 256 x 320 kernel into scratch spills)
    constexpr bool HEADS_FITS = WM * WN * NT * 2560 <= NST * STAGE_BYTES && MT * NT <= 8;
    if constexpr (HEADS_FITS) if (p.epi == EPI_HEADS && p.staged_epi && (p.rows_per_batch & 31) == 0 && (p.part_width & 31) == 0 &&
        (p.head_dim & 7) == 0 && (p.N & 31) == 0 && (p.M & 31) == 0) {
        igemm_epilogue_heads_staged<MT, NT, GLDS>(p, acc, mw0, nw0, lane, smem + wid * (NT * 2560), par);
        tl_end(p.tl);
        return;
    }
    igemm_epilogue<MT, NT, GLDS>(p, acc, mw0, nw0, lane, par);
    tl_end(p.tl);
}

// Finishes the K-split tiles: block (t, ij) sums the ksplit fp32 partials of ONE 32x32 MFMA sub-tile
// per wave (same lane/register geometry as the producing kernel) and runs the shared epilogue on it.
// grid = (tiles, MT*NT): 4x more blocks than tiles and 16 x ksplit independent loads per thread, so the
// pass is bandwidth- rather than latency-bound.
template <int WM, int WN, int WTM, int WTN>
__global__ void __launch_bounds__(64 * WM * WN)
igemm_reduce_kernel(const IGemmArgs p) {
    constexpr int NTHR = 64 * WM * WN;
    constexpr int BM = WM * WTM, BN = WN * WTN;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    const int ntn = (p.N + BN - 1) / BN;
    const int wg = p.n_main + blockIdx.x;
    const int ij = blockIdx.y, i = ij / NT, j = ij - i * NT;
    const int tile_m = wg / ntn, tile_n = wg - tile_m * ntn;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const float* base = p.ws + ((long)blockIdx.x * p.ksplit * (MT * NT * 16) + (long)ij * 16) * NTHR + tid;
    const long sstride = (long)(MT * NT * 16) * NTHR;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    int sidx = 0;
    for (; sidx + 2 <= p.ksplit; sidx += 2) {
        float v0[16], v1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { v0[r] = base[sidx * sstride + (long)r * NTHR]; v1[r] = base[(sidx + 1) * sstride + (long)r * NTHR]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += v0[r] + v1[r];
    }
    if (sidx < p.ksplit) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += base[sidx * sstride + (long)r * NTHR];
    }
    Par par; par.lds = nullptr; par.n0 = 0; par.b0 = 0; par.bnp = 0;      // parameters from global memory
    igemm_epilogue<1, 1, false>(p, acc, tile_m * BM + wm * WTM + i * 32, tile_n * BN + wn * WTN + j * 32, lane, par);
}

// ---- the same GEMM on 32-deep K-tiles (tile32_kernel) --------------------------------------------------------------------
// A K-tile of 64 fixes the LDS budget of the kernels above at (BM + BN) * 128 bytes per stage: the 256-wide tiles get TWO stages
// (one tile of lookahead - inside a forward shorter than the memory latency, which is why three stages took over every shape
// they fit) and one workgroup per CU.  Here a K-tile is 32 deep - LDS rows of 64 bytes, (BM + BN) * 64 bytes per
// stage - which buys, for the same LDS:
//   * 256 x 256 / 256 x 320 on a FOUR-stage ring (128 / 147 KB): three half-tiles of lookahead, requested and awaited at twice
//     the granularity (configs 13 / 20);
//   * 256 x 128 (8 waves, 72 KB), 128 x 128 and 256 x 64 (4 waves, 48 / 60 KB) on three stages with TWO or THREE workgroups per
//     CU (configs 15 / 16 / 17): one workgroup's prologue / epilogue runs under another's MFMAs and the chip's store bursts
//     de-synchronise - what the K <= 1280 projections of the transformer blocks lacked (round-3 timelines: a workgroup is
//     prologue 2 - 4 us -> 5 - 20 K-tiles -> epilogue 5 - 15 us with every CU bursting its stores at once).
// Layout: LDS-DMA piece = 16 rows x 64 B, lane -> (row = lane >> 2, 16-byte chunk = lane & 3); swizzle physical = logical ^
// ((row >> 2) & 3) applied to the DMA's SOURCE chunk; a fragment read (rows base + (lane & 31), logical chunk 2 * kstep +
// (lane >> 5)) touches, per lane group of a ds_read_b128, 16 rows whose (row & 3, (row >> 2) & 3) pairs are all different:
// conflict free (tests/test_lds_layout_cpu.py).  The A gather is igemm_kernel's (K-tile kt32 = half (kt32 & 1) of the 64-channel
// block / tap of kt32 >> 1); piece g = wave + NW * q of a K-tile is an activation piece for g < BM / 16, else a weight piece;
// when the pieces do not divide over the waves the first (A_P + B_P) % NW waves carry one more, and every counted wait uses the
// wave's own count.  One barrier per K-tile.  k is summed in igemm_kernel's order (16-deep MFMA steps, ascending): bit-identical
// results, so the in-situ tuner may pin these.  Whole K per tile (no K-split; ragged M / N edges are clamped and masked like
// igemm_kernel's); epilogues shared with igemm_kernel.
template <int WM, int WN, int WTM, int WTN, int NST, int WPE, int AMODE>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
tile32_kernel(const IGemmArgs p) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * WTM, BN = WN * WTN;
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int A_P = BM / 16, B_P = BN / 16, TOT = A_P + B_P;      // 16-row DMA pieces of the two operand tiles
    constexpr int FULL = TOT / NW, REM = TOT % NW, PPW = FULL + (REM ? 1 : 0);
    constexpr int NM = 2 * MT * NT;                        // MFMAs per wave per K-tile
    constexpr int STAGE_BYTES = (BM + BN) * 64;
    constexpr int PAR_OFF = NST * STAGE_BYTES;             // epilogue parameters behind the ring
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tl_begin(p.tl);

    const int bid = blockIdx.x;
    int wg;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    tile_of(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid - wm * WN;
    const bool extra = REM != 0 && wid < REM;              // (wave-uniform) this wave carries PPW pieces, the others PPW - 1
    const int HW = p.rows_per_batch;

    // ---- loader state: per piece slot q, either an activation row (a_pix, gathered per K-tile) or a weight-row pointer ----
    const int prow = lane >> 2, pchunk = lane & 3;
    const int schunk = pchunk ^ ((prow >> 2) & 3);         // the logical chunk that belongs at physical slot pchunk of that row
    int a_pix[PPW];
    const half_t* b_ptr[PPW];
    int dst[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int g = wid + NW * q;
        a_pix[q] = 0; b_ptr[q] = p.w;
        if (g < A_P) {
            int m = m0 + g * 16 + prow;
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
            dst[q] = g * 16 * 64;
        } else {
            int n = n0 + (g - A_P) * 16 + prow;
            n = n < p.N ? n : p.N - 1;
            b_ptr[q] = p.w + (long)n * p.K + schunk * 8;
            dst[q] = BM * 64 + (g - A_P) * 16 * 64;
        }
    }
    struct Gather { const half_t* src; int cs, Cs, dpix, dy, dx; };
    auto gather_of = [&](int kt) {                          // kt = 32-deep K-tile
        Gather g;
        const int k64 = kt >> 1;
        int tap = 0, cc = k64 << 6;
        if (p.taps == 9) { const int cb = k64 / 9; tap = k64 - cb * 9; cc = cb << 6; }
        const bool s0 = cc < p.C0;
        g.src = s0 ? p.a0 : p.a1;
        g.cs = (s0 ? cc : cc - p.C0) + ((kt & 1) << 5); g.Cs = s0 ? p.C0 : p.C1;
        g.dy = 0; g.dx = 0;
        if (p.taps == 9) { g.dy = tap / 3 - 1; g.dx = tap - (tap / 3) * 3 - 1; }
        g.dpix = 0;
        if constexpr (AMODE == 1) g.dpix = g.dy * (p.W + 2) + g.dx;
        else if constexpr (AMODE == 2) g.dpix = g.dy * (2 * p.W + 2) + g.dx;
        return g;
    };
    auto piece = [&](int q, int kt, int stage, const Gather& g) {      // q compile-time after unrolling
        char* base = smem + stage * STAGE_BYTES;
        const int gi = wid + NW * q;
        if (gi < A_P) {
            int pix;
            if constexpr (AMODE == 3) {
                const int b = a_pix[q] >> 22, y = (a_pix[q] >> 11) & 2047, x = a_pix[q] & 2047;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                pix = (b * (Hs + 2) + ((y + g.dy) >> 1) + 1) * (Ws + 2) + ((x + g.dx) >> 1) + 1;
            } else {
                pix = a_pix[q] + g.dpix;
            }
            const half_t* gp = g.src + (long)pix * g.Cs + g.cs + schunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(base + dst[q]), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[q] + ((long)kt << 5)),
                                             (__attribute__((address_space(3))) void*)(base + dst[q]), 16, 0, 0);
        }
    };
    auto dma_tile = [&](int kt, int stage) {               // all of the wave's pieces back to back (prologue)
        const Gather g = gather_of(kt);
#pragma unroll
        for (int q = 0; q < PPW; ++q) if (q < FULL || extra) piece(q, kt, stage, g);
    };
    // "at most n K-tiles of THIS wave's pieces still in flight"
#define CFGPP_T32_WAIT(n) do { if (REM != 0 && extra) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * PPW) : "memory");          \
                               else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * FULL) : "memory"); } while (0)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int frow = lane & 31, fhi = lane >> 5;
    const int fsw = (frow >> 2) & 3;
    const int a_rd = (wm * WTM + frow) * 64;
    const int b_rd = BM * 64 + (wn * WTN + frow) * 64;

    const int nk = p.K >> 5;
    tl_stamp(p.tl, 8);
    par_stage<BN, NW>(p, smem + PAR_OFF, n0, m0, wid, lane);      // oldest loads of the kernel: covered by every counted wait
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_) if (s_ < nk) dma_tile(s_, s_);
    tl_stamp(p.tl, 9);

    // one K-tile: MFMAs on `stage`; with_dma: the wave's pieces of tile ktn go out into stage `nxt`, one after every second MFMA
    auto tile_body = [&](int stage, int ktn, int nxt, auto with_dma) {
        constexpr bool DMA = decltype(with_dma)::value;
        const char* S = smem + stage * STAGE_BYTES;
        Gather gn = {p.a0, 0, p.C0, 0, 0, 0};
        if constexpr (DMA) gn = gather_of(ktn);
        int issued = 0, done = 0;                          // compile-time after unrolling
        auto after_mfma = [&]() {
            ++done;
            if constexpr (DMA) {
                if (issued < PPW && done >= 2 * issued + 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (issued < FULL || extra) piece(issued, ktn, nxt, gn);
                    ++issued;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if constexpr (MT * NT <= 8) {
            half8_t xa[2][MT], wb[2][NT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = (((ks << 1) | fhi) ^ fsw) << 4;
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[ks][i] = *reinterpret_cast<const half8_t*>(S + a_rd + i * 32 * 64 + coff);
#pragma unroll
                for (int j = 0; j < NT; ++j) wb[ks][j] = *reinterpret_cast<const half8_t*>(S + b_rd + j * 32 * 64 + coff);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[ks][j], xa[ks][i], acc[i][j], 0, 0, 0);
                        after_mfma();
                    }
        } else {
            // 10 accumulator tiles per wave (256 x 320): one fragment set at a time
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int coff = (((ks << 1) | fhi) ^ fsw) << 4;
                half8_t xa[MT], wb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) xa[i] = *reinterpret_cast<const half8_t*>(S + a_rd + i * 32 * 64 + coff);
#pragma unroll
                for (int j = 0; j < NT; ++j) wb[j] = *reinterpret_cast<const half8_t*>(S + b_rd + j * 32 * 64 + coff);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                        after_mfma();
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (DMA) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) if (q >= issued && (q < FULL || extra)) piece(q, ktn, nxt, gn);      // (what did not fit between the MFMAs)
        }
    };
    // Iteration kt: "tile kt has landed" (at most the NST-2 younger tiles' pieces outstanding) + barrier - which also says every
    // wave is done reading tile kt-1, whose stage the pieces of tile kt+NST-1 (issued during this tile's MFMAs) overwrite.
    int kt = 0, cur = 0, nxt = NST - 1;                  // stage of tile kt / of tile kt + NST - 1
    for (; kt + NST - 1 < nk; ++kt) {
        CFGPP_T32_WAIT(NST - 2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt == 0) tl_stamp(p.tl, 1);
        tile_body(cur, kt + NST - 1, nxt, std::true_type{});
        if (kt == 0) tl_stamp(p.tl, 7);
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    for (; kt < nk; ++kt) {                              // no tile left to request: drain
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tile_body(cur, 0, 0, std::false_type{});
        cur = cur + 1 == NST ? 0 : cur + 1;
    }
#undef CFGPP_T32_WAIT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                     // every wave is done with the ring: LDS is free for the epilogue's staging
    tl_stamp(p.tl, 2);

    const int mw0 = m0 + wm * WTM, nw0 = n0 + wn * WTN;
    Par par;
    par.lds = smem + PAR_OFF; par.n0 = n0; par.b0 = HW > 0 ? qdiv(m0, HW) : 0; par.bnp = par_bnp(BN);
    constexpr bool STAGED_FITS = NW * 32 * (WTN * 2 + 16) <= NST * STAGE_BYTES;
    if (STAGED_FITS && p.epi == EPI_STORE && (p.N & 7) == 0 && p.staged_epi) {
        igemm_epilogue_staged<MT, NT, true>(p, acc, mw0, nw0, lane, smem + wid * (32 * (WTN * 2 + 16)), par);
        tl_end(p.tl);
        return;
    }
    if constexpr (NT % 2 == 0) {
        if (p.epi == EPI_GEGLU && (p.N & 127) == 0 && p.staged_epi && p.omode == 0) {
            igemm_epilogue_geglu_staged<MT, NT, true>(p, acc, mw0, nw0, lane, smem + wid * (32 * ((NT / 2) * 64 + 16)), par);
            tl_end(p.tl);
            return;
        }
    }
    constexpr bool HEADS_FITS = NW * NT * 2560 <= NST * STAGE_BYTES && MT * NT <= 8;
    if constexpr (HEADS_FITS) if (p.epi == EPI_HEADS && p.staged_epi && (p.rows_per_batch & 31) == 0 && (p.part_width & 31) == 0 &&
        (p.head_dim & 7) == 0 && (p.N & 31) == 0 && (p.M & 31) == 0) {
        igemm_epilogue_heads_staged<MT, NT, true>(p, acc, mw0, nw0, lane, smem + wid * (NT * 2560), par);
        tl_end(p.tl);
        return;
    }
    igemm_epilogue<MT, NT, true>(p, acc, mw0, nw0, lane, par);
    tl_end(p.tl);
}

// ---- 128 x 160 tile as EIGHT waves of 32 x 80 on v_mfma_f32_16x16x32_f16 --------------------------------------------
// The in-situ A/Bs of round 2 rewarded three properties at once - 8 waves per workgroup (two per SIMD), >= 3 LDS stages
// (two K-tiles of lookahead: inside a forward the operands come from HBM / the Infinity Cache) and one tile per CU - and no
// 32 x 32 tiling has all three for the M = 4096 x N = 1280 class (half of an SDXL forward, the 16x16 level of SD1.5): 128 x 160
// is 256 tiles but 32 x 160 waves make it a 4-wave workgroup, 256 x 128 on 8 waves is 160 tiles.  80 = 5 x 16, so the
// 16 x 16 x 32 MFMA gives each of 8 waves a 32 x 80 slab (2 x 5 accumulator tiles of 4 registers).
//   * operands "swapped" as in igemm_kernel: MFMA-A = weights (rows n), MFMA-B = activations (columns m); a lane then holds
//     4 consecutive output features (n = 4 * (lane >> 4) + r) of ONE pixel (m = lane & 15);
//   * LDS image, XOR swizzle and the LDS-DMA loader are igemm_kernel's; a fragment is rows base + (lane & 15), logical
//     16-byte chunk 4 * s + (lane >> 4) of k-step s (two k-steps of 32 per K-tile): with 16-row-aligned bases the four
//     16-lane groups of a ds_read_b128 hit 16 distinct bank quads (same argument as the 32-row case);
//   * 8 waves x 8 rows = 64 rows per loader pass: 128 activation rows = 2 passes, 160 weight rows = 2 passes + one that
//     only waves 0-3 take, so the counted vmcnt waits use the WAVE's piece count (5 or 4 per K-tile);
//   * plain-store epilogue (bias / temb / residual through the per-wave LDS transpose) and a head-major one
//     (cfgpp_igemm_set_mf16_heads); whole tiles only (no K-split).
// The 16 x 16 x 32 MFMA sums k in a different order than the 32 x 32 x 16 one, so this tile is NOT a tuner candidate (the
// tuner's choices must not change results): it is used by rule (igemm_launch; cfgpp_igemm_set_mf16) or forced (configs 18 / 19).
template <bool L = true>
__device__ __forceinline__ void igemm_epilogue_staged16(const IGemmArgs& p, f32x4 (&acc)[2][5], int mw0, int nw0, int lane,
                                                        char* stg /* wave-private, 32 * 176 bytes */, const Par& par) {
    constexpr int WTN = 80, PITCH = WTN * 2 + 16, CPR = WTN / 8, NQ = (32 * CPR + 63) / 64;
    const int c16 = lane & 15, fq = lane >> 4;
    const int HW = p.rows_per_batch;
    // rows leave as 16-byte pieces; lane r (< 32) knows row r's output / residual pixel index
    const int frow = lane & 31;
    const int mr = mw0 + frow;
    const int mrc = mr < p.M ? mr : p.M - 1;
    int opix = mrc, rpix = mrc;
    if (p.omode == 1 || p.rmode == 1) {
        const int pp = padded_pix(mrc, HW, p.W, p.H);
        if (p.omode == 1) opix = pp;
        if (p.rmode == 1) rpix = pp;
    }
    // Order matters.  (1) the parameter segments (bias + time embedding of this lane's columns and rows) are read from LDS FIRST:
    // each of those reads carries a compiler-inserted vmcnt(0) (the segments were written by LDS-DMA), free while nothing else
    // is in flight.  (2) the residual pieces are requested in ONE block of loads (clamped addresses, no per-piece branch).
    // (3) scale / add / convert / transpose through LDS run under their latency.  (4) read back, add the residual, store.
    float4 add[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = mw0 + i * 16 + c16;
        const int mc = m < p.M ? m : p.M - 1;
        const int b = (HW > 0) ? qdiv(mc, HW) : 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            int n = nw0 + j * 16 + 4 * fq;
            n = n < p.N ? n : p.N - 4;
            add[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) add[i][j] = par_bias4<L>(p, par, n);
            if (p.temb) {
                const float4 tt = par_temb4<L>(p, par, b, n);
                add[i][j].x += tt.x; add[i][j].y += tt.y; add[i][j].z += tt.z; add[i][j].w += tt.w;
            }
        }
    }
    half8_t rres[NQ];
    if (p.resid) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = lane + 64 * q;
            const int r = c / CPR, cc = c - r * CPR;
            const int rp = __shfl(rpix, r);
            int n = nw0 + cc * 8;
            n = n < p.N ? n : p.N - 8;
            rres[q] = *reinterpret_cast<const half8_t*>(p.resid + (long)rp * p.rld + n);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int nl = j * 16 + 4 * fq;
            half4_t o;
            o[0] = (half_t)(acc[i][j][0] * p.out_scale + add[i][j].x);
            o[1] = (half_t)(acc[i][j][1] * p.out_scale + add[i][j].y);
            o[2] = (half_t)(acc[i][j][2] * p.out_scale + add[i][j].z);
            o[3] = (half_t)(acc[i][j][3] * p.out_scale + add[i][j].w);
            *reinterpret_cast<half4_t*>(stg + (i * 16 + c16) * PITCH + nl * 2) = o;
        }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = lane + 64 * q;
        const int r = c / CPR, cc = c - r * CPR;
        const int op = __shfl(opix, r);
        const int mm = mw0 + r, n = nw0 + cc * 8;
        half8_t v = *reinterpret_cast<const half8_t*>(stg + (r & 31) * PITCH + cc * 16);
        if (p.resid) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)rres[q][k]);
            if (p.gstat && c < 32 * CPR) *reinterpret_cast<half8_t*>(stg + (r & 31) * PITCH + cc * 16) = v;
        }
        if (c < 32 * CPR && mm < p.M && n < p.N) *reinterpret_cast<half8_t*>(p.out + (long)op * p.old + n) = v;
    }
    if (p.gstat) gstat_block<WTN, PITCH>(p, stg, mw0, nw0, lane);
}

// EPI_HEADS for the 16 x 16 accumulator layout: the wave's 32-token x 80-column slab goes through LDS as five
// 32-token x 16-column blocks - row-major [token][column] (48-byte pitch) for Q / K columns, transposed
// [head dim][key position] (80-byte pitch, positions in the attention kernel's permuted order unless vt_linear) for V
// columns - and leaves as 16-byte pieces: 8 head dims of one token, or 8 key positions of one head dim.
// Needs rows_per_batch % 32 == 0, part_width % 16 == 0, head_dim % 8 == 0 (checked by mf16_supports).
template <bool L = true>
__device__ __forceinline__ void igemm_epilogue_heads_staged16(const IGemmArgs& p, f32x4 (&acc)[2][5], int mw0, int nw0, int lane,
                                                              char* stg /* wave-private, 5 * 1536 bytes */, const Par& par) {
    constexpr int BLK = 1536, QK_PITCH = 48, VT_PITCH = 80;
    if (mw0 >= p.M) return;
    const int c16 = lane & 15, fq = lane >> 4;
    const int HW = p.rows_per_batch;
    const int mw0u = __builtin_amdgcn_readfirstlane(mw0);
    const int b = qdiv(mw0u, HW), tok0 = mw0u - b * HW;            // one batch, one aligned 32-token block
    HeadCol hc[5];                                             // per 16-column group: part / head / offset of its first column
#pragma unroll
    for (int j = 0; j < 5; ++j) hc[j] = head_col(p, nw0 + j * 16);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int ng = nw0 + j * 16;                           // first column of the 16-column group (ng < N: N % 160 == 0)
        const int part = hc[j].part;
        char* blk = stg + j * BLK;
        const int n = ng + 4 * fq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = i * 16 + c16;                        // token row inside the 32-token block
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[i][j][k];
            if (p.bias) {
                const float4 bb = par_bias4<L>(p, par, n);
                v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
            }
            if (part == 2) {
                const int pos = p.vt_linear ? r : cfgpp_vt_pos(r);
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<half_t*>(blk + (4 * fq + k) * VT_PITCH + pos * 2) = (half_t)v[k];
            } else {
                half4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                *reinterpret_cast<half4_t*>(blk + r * QK_PITCH + (4 * fq) * 2) = o;
            }
        }
    }
    // 64 pieces of 16 bytes per group: one per lane
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int part = hc[j].part;
        const char* blk = stg + j * BLK;
        if (part == 2) {                                       // row = head-dim column ng + r, piece = 8 key positions
            const int r = lane >> 2, c4 = lane & 3;
            const half8_t v = *reinterpret_cast<const half8_t*>(blk + r * VT_PITCH + c4 * 16);
            int head, dd;
            head_step(p, hc[j], r, head, dd);
            const long bh = (long)b * p.heads + head;
            *reinterpret_cast<half8_t*>(p.hvt + (bh * p.head_dim_pad + dd) * p.tok_pad + tok0 + 8 * c4) = v;
        } else {                                               // row = token tok0 + r, piece = 8 head dims
            const int r = lane >> 1, c2 = lane & 1;
            const half8_t v = *reinterpret_cast<const half8_t*>(blk + r * QK_PITCH + c2 * 16);
            int head, dd;
            head_step(p, hc[j], 8 * c2, head, dd);
            const long bh = (long)b * p.heads + head;
            half_t* base = part == 0 ? p.hq : p.hk;
            const int tp = part == 0 ? p.q_tok_pad : p.tok_pad;
            *reinterpret_cast<half8_t*>(base + (bh * tp + tok0 + r) * p.head_dim_pad + dd) = v;
        }
    }
}

template <int AMODE, int NST>
__global__ void __launch_bounds__(512)
igemm16_kernel(const IGemmArgs p) {
    constexpr int BM = 128, BN = 160, RSTEP = 64;
    constexpr int A_CH = BM / RSTEP;                // 2 loader passes over the activation rows
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int NM = 20;                          // MFMAs per wave per K-tile: 2 k-steps x (2 x 5) tiles
    static_assert(NST >= 3 && NST <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tl_begin(p.tl);

    const int bid = blockIdx.x;
    int wg;
    {
        const int nwg = gridDim.x;
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
    const bool b3 = wid < 4;                        // (wave-uniform) this wave also loads weight rows 128 + 8 wid ..

    // ---- loader state (as igemm_kernel) ----
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int HW = p.rows_per_batch;
    int a_pix[A_CH];
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
        int m = m0 + lrow + j * RSTEP;
        m = m < p.M ? m : p.M - 1;
        if constexpr (AMODE == 0) a_pix[j] = m;
        else if constexpr (AMODE == 1) a_pix[j] = padded_pix(m, HW, p.W, p.H);
        else if constexpr (AMODE == 2) {
            const int b = qdiv(m, HW), q = m - b * HW, y = qdiv(q, p.W), x = q - y * p.W;
            a_pix[j] = (b * (2 * p.H + 2) + 2 * y + 1 + p.ashift) * (2 * p.W + 2) + 2 * x + 1 + p.ashift;
        } else {
            const int b = qdiv(m, HW), q = m - b * HW, y = qdiv(q, p.W), x = q - y * p.W;
            a_pix[j] = (b << 22) | (y << 11) | x;
        }
    }
    const int schunk = lchunk ^ ((lrow >> 1) & 7);  // the DMA stores lane-linear: swizzle the SOURCE chunk
    const half_t* b_ptr[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int n = n0 + lrow + j * RSTEP;
        n = n < p.N ? n : p.N - 1;
        b_ptr[j] = p.w + (long)n * p.K + schunk * 8;
    }
    const int wave_row0 = wid * 8;

    struct Gather { const half_t* src; int cs, Cs, dpix, dy, dx; };
    auto gather_of = [&](int kt) {
        Gather g;
        int tap = 0, cc = kt << 6;
        if (p.taps == 9) { const int cb = kt / 9; tap = kt - cb * 9; cc = cb << 6; }
        const bool s0 = cc < p.C0;
        g.src = s0 ? p.a0 : p.a1;
        g.cs = s0 ? cc : cc - p.C0; g.Cs = s0 ? p.C0 : p.C1;
        g.dy = 0; g.dx = 0;
        if (p.taps == 9) { g.dy = tap / 3 - 1; g.dx = tap - (tap / 3) * 3 - 1; }
        g.dpix = 0;
        if constexpr (AMODE == 1) g.dpix = g.dy * (p.W + 2) + g.dx;
        else if constexpr (AMODE == 2) g.dpix = g.dy * (2 * p.W + 2) + g.dx;
        return g;
    };
    // piece q of K-tile kt into `stage`: q = 0, 1 activation passes; 2, 3 weight passes; 4 the half weight pass (b3 waves)
    auto piece = [&](int q, int kt, int stage, const Gather& g) {
        char* As = smem + stage * STAGE_BYTES;
        char* Bs = As + BM * 128;
        if (q < A_CH) {
            const int j = q;
            int pix;
            if constexpr (AMODE == 3) {
                const int b = a_pix[j] >> 22, y = (a_pix[j] >> 11) & 2047, x = a_pix[j] & 2047;
                const int Hs = p.H >> 1, Ws = p.W >> 1;
                pix = (b * (Hs + 2) + ((y + g.dy) >> 1) + 1) * (Ws + 2) + ((x + g.dx) >> 1) + 1;
            } else {
                pix = a_pix[j] + g.dpix;
            }
            const half_t* gp = g.src + (long)pix * g.Cs + g.cs + schunk * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(As + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
        } else {
            const int j = q - A_CH;
            const half_t* gp = b_ptr[j] + ((long)kt << 6);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(Bs + (wave_row0 + j * RSTEP) * 128), 16, 0, 0);
        }
    };
    // "at most n K-tiles of THIS wave's pieces still in flight"
#define CFGPP_WAIT_TILES(n) do { if (b3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * 5) : "memory");          \
                                 else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * 4) : "memory"); } while (0)

    f32x4 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][j][k] = 0.f;

    // fragment reads: row = base + (lane & 15), logical chunk = 4 * s + (lane >> 4)
    const int c16 = lane & 15, fq = lane >> 4;
    const int fsw = c16 >> 1;
    const int a_rd = (wm * 32 + c16) * 128;
    const int b_rd = (wn * 80 + c16) * 128;

    const int nk = p.K >> 6;
    tl_stamp(p.tl, 8);
    constexpr int PAR_OFF = NST * STAGE_BYTES;      // epilogue parameters behind the ring
    par_stage<BN, 8>(p, smem + PAR_OFF, n0, m0, wid, lane);          // oldest loads of the kernel: covered by every counted wait
#pragma unroll
    for (int s_ = 0; s_ < NST - 1; ++s_)
        if (s_ < nk) {
            const Gather g = gather_of(s_);
#pragma unroll
            for (int q = 0; q < 4; ++q) piece(q, s_, s_, g);
            if (b3) piece(4, s_, s_, g);
        }
    tl_stamp(p.tl, 9);
    if (nk >= NST - 1) CFGPP_WAIT_TILES(NST - 2);      // NST-1 tiles issued: the oldest has landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    tl_stamp(p.tl, 1);

    // K-tile schedule (round 3; the timeline of round 2's loop showed ~1300-1450 cycles per K-tile against an MFMA floor of
    // 644: after every end-of-tile barrier all 8 waves first read BOTH k-steps' fragments (112 KB through the LDS pipe,
    // nothing for the matrix pipe to do), and the two waves of a SIMD reached their LDS-DMA instructions - ~170 cycles of
    // issue each at this rate - together):
    //   * the barrier sits in the MIDDLE of a tile.  Phase 0 = the k-step-0 MFMAs of tile kt, whose fragments were read
    //     during phase 1 of tile kt-1; phase 1 = the k-step-1 MFMAs, fragments read during phase 0.  Every fragment read has
    //     ten MFMAs (~340 cycles) to land and is issued right AFTER the first MFMA of a phase, so the wait the compiler
    //     puts in front of that MFMA cannot cover it.
    //   * "tile kt+1 has landed" is waited for before that mid-tile barrier; the k-step-0 fragments of tile kt+1 are read
    //     behind it.  A stage is free again once the barrier after its last reads has been passed, which is still before
    //     the first DMA into it.
    //   * the waves 0-3 (one per SIMD) issue their DMA pieces of tile kt+NST-1 during phase 0, their SIMD partners 4-7
    //     during phase 1: while one wave of a SIMD sits in the memory pipe's queue the other one feeds the matrix pipe.
    //     (waves 4-7 therefore have one tile less in flight at the barrier: their counted wait is one tile shorter.)
    half8_t xa[2][2], wb[2][5];
    auto rd_frags = [&](int s, int stage) {
        const char* As = smem + stage * STAGE_BYTES;
        const char* Bs = As + BM * 128;
        const int coff = (((s << 2) | fq) ^ fsw) << 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) xa[s][i] = *reinterpret_cast<const half8_t*>(As + a_rd + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < 5; ++j) wb[s][j] = *reinterpret_cast<const half8_t*>(Bs + b_rd + j * 16 * 128 + coff);
    };
    rd_frags(0, 0);

    auto tile_body = [&](int kt, int cur, int ktn, int nxt, auto with_dma) {
        constexpr bool DMA = decltype(with_dma)::value;
        Gather gn = {p.a0, 0, p.C0, 0, 0, 0};
        if constexpr (DMA) gn = gather_of(ktn);
        // phase s: ten MFMAs; after the first one the fragments of the NEXT phase are requested; the wave's DMA pieces go out
        // after every second MFMA of ITS phase
        auto phase = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            const bool mine = (s == 0) == b3;
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const int i = m / 5, j = m - 5 * (m / 5);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[s][j], xa[s][i], acc[i][j], 0, 0, 0);
                if (m == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s == 0) rd_frags(1, cur);
                    else if (kt + 1 < nk) rd_frags(0, (kt + 1) % NST);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (DMA) {
                    if ((m & 1) == 1) {
                        const int q = m >> 1;                        // 0 .. 4
                        __builtin_amdgcn_sched_barrier(0);
                        if (mine && (q < 4 || b3)) piece(q, ktn, nxt, gn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        phase(std::integral_constant<int, 0>{});
        // tile kt+1 has landed: waves 0-3 have issued up to tile kt+NST-1 (NST-2 younger tiles may be in flight), waves 4-7 up to
        // kt+NST-2 (NST-3)
        if constexpr (DMA) {
            if (b3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * 5) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 3) * 4) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        phase(std::integral_constant<int, 1>{});
    };
    int kt = 0;
    for (; kt + NST - 1 < nk; ++kt) {
        tile_body(kt, kt % NST, kt + NST - 1, (kt + NST - 1) % NST, std::true_type{});
        if (kt == 0) tl_stamp(p.tl, 7);
    }
    for (; kt < nk; ++kt) tile_body(kt, kt % NST, 0, 0, std::false_type{});
    // (the last mid-tile barrier was passed with vmcnt(0) and every LDS read complete: LDS is free for the epilogue)
    tl_stamp(p.tl, 2);
#undef CFGPP_WAIT_TILES

    Par par;
    par.lds = smem + PAR_OFF; par.n0 = n0; par.b0 = HW > 0 ? qdiv(m0, HW) : 0; par.bnp = par_bnp(BN);
    if (p.epi == EPI_HEADS) igemm_epilogue_heads_staged16(p, acc, m0 + wm * 32, n0 + wn * 80, lane, smem + wid * (5 * 1536), par);
    else igemm_epilogue_staged16(p, acc, m0 + wm * 32, n0 + wn * 80, lane, smem + wid * (32 * 176), par);
    tl_end(p.tl);
}

// ---- launch + tail scheduling ------------------------------------------------------------------
// diagnostics: per-workgroup timeline of the `target`-th igemm_launch call after the (re)arming call
static unsigned long long* g_tl = nullptr; static long g_tl_cap = 0; static int g_tl_target = -1, g_tl_count = 0;
static int g_tl_info[12] = {0};                     // {cfg, grid, threads, BM, BN, NST, ksplit, n_major, M, N, K, epi} of the recorded launch
static unsigned long long* tl_take(int cfg, int grid, int threads, int BM, int BN, int NST, const IGemmArgs& a) {
    if (!a.tl || grid > g_tl_cap) return nullptr;
    const int info[12] = {cfg, grid, threads, BM, BN, NST, a.ksplit, a.n_major, a.M, a.N, a.K, a.epi};
    for (int i = 0; i < 12; ++i) g_tl_info[i] = info[i];
    return g_tl;
}
// fp32 partial workspace of the K-split launches.  Engines bring their own (IGemmArgs::ws_buf: allocated when the plan is
// built, so no hipMalloc - an implicit device sync, and illegal inside a stream capture - can happen mid-forward, and freed with
// the engine).  Launches without one (the single-op test wrappers) share a process-global buffer PER STREAM (one device per process).  Launches of one stream are
// serialised, so they can share a buffer; two engines driven on different streams (UNet and VAE, or two UNets) may have
// K-split launches in flight at the same time, and those must not share partials.
constexpr long WS_BYTES = 128L << 20;               // fp32 partials of one launch: T * S * BM * BN * 4 bytes must fit
static std::map<hipStream_t, float*> g_ws_by_stream;
static float* ws_for(hipStream_t stream) {
    auto it = g_ws_by_stream.find(stream);
    if (it != g_ws_by_stream.end()) return it->second;
    float* p = nullptr;
    if (hipMalloc((void**)&p, (size_t)WS_BYTES) != hipSuccess) p = nullptr;
    if (p) g_ws_by_stream[stream] = p;
    return p;
}
static int g_staged_epi = 1;
static int g_last_hint_applied = 1;                 // see igemm_last_hint_applied()
extern "C" void cfgpp_igemm_set_staged_epilogue(int on) { g_staged_epi = on ? 1 : 0; }
static int g_big_tiles = 1;
extern "C" void cfgpp_igemm_set_big_tiles(int on) { g_big_tiles = on ? 1 : 0; }
static int g_tail_split = 1;                        // 1 = K-split tiny grids with long K (8x8-level convs)
static int g_n_major = -1;                          // tile walk: -1 = by operand bytes, 0 = always M-major, 1 = always N-major
extern "C" void cfgpp_igemm_set_n_major(int mode) { g_n_major = mode; }

// time-embedding rows a BM-row tile stages: every batch its rows can touch (a tile starts anywhere inside a batch)
static int par_slots(const IGemmArgs& a, int BM) {
    if (a.temb == nullptr || a.rows_per_batch <= 0) return 0;
    const int HW = a.rows_per_batch, nbatch = cdiv(a.M, HW);
    const int span = (BM + HW - 1) / HW + 1;
    return span < nbatch ? span : nbatch;
}

// ---- XCD-blocked 2-D tile walk (IGemmArgs::walk_bn) ----------------------------------------------------------------------
// An XCD (32 CUs, one 4-MiB L2) runs its share of the tiles in ROUNDS of the workgroups resident at a time, and what a round pulls
// through the fabric is the set of distinct activation row-blocks and weight slabs its tiles touch.  With a 1-D walk a round of 32
// tiles is a strip - e.g. the SDXL GEGLU (16 x 32 tiles of 256 x 320), N-major: every round touches ALL 16 activation row-blocks
// (10.5 MB, more than the L2 holds) beside 2 weight slabs, so the activation matrix crosses the fabric twice per XCD; the PMC
// passes folded onto launch classes (profiles/r05/pmc_per_launch_class_*.txt) show exactly that count, 3.0x the algorithmic bytes.
// A rectangular block per XCD makes the rounds rectangles too.  walk_plan() counts the bytes per XCD for the walk the launch would
// take and for the 2 x 4 / 4 x 2 block forms (either inner order) and switches when a block form saves >= 15 %.  Results do not
// depend on the walk; the tuner's pinned 1-D walks are left alone.
static int g_walk_blocked = 1;
extern "C" void cfgpp_igemm_set_blocked_walk(int on) { g_walk_blocked = on ? 1 : 0; }
static void walk_plan(IGemmArgs& a, int BM, int BN, int ntm, int ntn, int smem) {
    a.walk_bn = 0; a.walk_per = 0; a.walk_tmb = 0; a.walk_tnb = 0;
    const int T = ntm * ntn;
    if (!g_walk_blocked || a.walk_hint || (T & 7) || T < 64 || a.ksplit > 1 || a.n_main != T) return;
    const int per = T >> 3;
    int wpc = smem > 0 ? (160 * 1024) / smem : 1;
    wpc = wpc < 1 ? 1 : wpc > 4 ? 4 : wpc;
    const int R = 32 * wpc;                                   // tiles an XCD runs at a time
    const double a_row = 2.0 * BM * (a.C0 + a.C1) * (a.amode == 2 ? 4.0 : a.amode == 3 ? 0.25 : 1.0);
    const double w_col = 2.0 * BN * a.K;
    // bytes XCD 0 pulls in, round by round: distinct tile rows x a_row + distinct tile columns x w_col
    auto cost = [&](auto tile_at) {
        double tot = 0.0;
        for (int r0 = 0; r0 < per; r0 += R) {
            unsigned long long seen_m[8] = {0}, seen_n[8] = {0};      // bitsets: ntm, ntn <= 512
            int nm = 0, nn = 0;
            for (int i = r0; i < per && i < r0 + R; ++i) {
                int tm, tn; tile_at(i, tm, tn);
                if (tm < 512 && !((seen_m[tm >> 6] >> (tm & 63)) & 1ull)) { seen_m[tm >> 6] |= 1ull << (tm & 63); ++nm; }
                if (tn < 512 && !((seen_n[tn >> 6] >> (tn & 63)) & 1ull)) { seen_n[tn >> 6] |= 1ull << (tn & 63); ++nn; }
            }
            tot += nm * a_row + nn * w_col;
        }
        return tot;
    };
    if (ntm > 512 || ntn > 512) return;
    const int div1 = a.n_major ? ntm : ntn;
    const double now = cost([&](int i, int& tm, int& tn) { const int q = i / div1, r = i - q * div1; tm = a.n_major ? r : q; tn = a.n_major ? q : r; });
    double best = now; int best_bn = 0, best_inner = 0, best_tmb = 0, best_tnb = 0;
    for (int bn : {4, 2}) {
        const int bm = 8 / bn;
        if (ntm % bm || ntn % bn) continue;
        const int tmb = ntm / bm, tnb = ntn / bn;
        for (int inner = 0; inner < 2; ++inner) {             // 0: M-major inside the block, 1: N-major
            const double c = cost([&](int i, int& tm, int& tn) {
                if (inner) { tn = i / tmb; tm = i - tn * tmb; } else { tm = i / tnb; tn = i - tm * tnb; } });
            if (c < best) { best = c; best_bn = bn; best_inner = inner; best_tmb = tmb; best_tnb = tnb; }
        }
    }
    if (best_bn && best <= 0.85 * now) {
        a.walk_bn = best_bn; a.walk_per = per; a.walk_tmb = best_tmb; a.walk_tnb = best_tnb; a.n_major = best_inner;
    }
}

// host-only probe of walk_plan for tests (no GPU involved): token GEMM M x N x K on BM x BN tiles with `smem` bytes of LDS per
// workgroup and the given 1-D default; out5 = {walk_bm (0 = the 1-D walk stays), walk_bn, tiles per block along M, along N, inner
// order (1 = N-major)}
extern "C" void cfgpp_igemm_walk_plan_probe(int M, int N, int K, int BM, int BN, int smem, int n_major, int* out5) {
    IGemmArgs a; std::memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.C0 = K; a.amode = 0; a.ksplit = 1; a.n_major = n_major;
    const int ntm = cdiv(M, BM), ntn = cdiv(N, BN);
    a.n_main = ntm * ntn;
    walk_plan(a, BM, BN, ntm, ntn, smem);
    out5[0] = a.walk_bn ? 8 / a.walk_bn : 0; out5[1] = a.walk_bn; out5[2] = a.walk_tmb; out5[3] = a.walk_tnb; out5[4] = a.n_major;
}

// GroupNorm statistics of the output (IGemmArgs::gstat): only the LDS-staged plain-store epilogues of whole-K tiles write them.
// Decided HERE, per launch, and reported to the caller through *stat_flag (host memory): a consumer must not trust a buffer the
// launch did not fill.
static void stats_decide(IGemmArgs& a, bool staged_store) {
    const bool on = a.gstat != nullptr && staged_store && a.epi == EPI_STORE && (a.N & 7) == 0 && (a.M & 31) == 0 && a.ksplit <= 1 && a.n_main > 0;
    if (!on) a.gstat = nullptr;
    if (a.stat_flag) *a.stat_flag = on ? 1 : 0;
}

template <int WM, int WN, int WTM, int WTN, bool GLDS, int AMODE, int NST = 2>
int launch_cfg_amode(const IGemmArgs& a_in, hipStream_t stream) {
    constexpr int BM = WM * WTM, BN = WN * WTN, NTHR = 64 * WM * WN;
    constexpr int smem_std = NST * (BM + BN) * 128 + (GLDS ? par_bytes(BN) : 0);      // ring + epilogue-parameter segments
    static_assert(smem_std <= 160 * 1024, "tile does not fit the LDS");
    constexpr int blocks_per_cu = (160 * 1024) / smem_std < 8 ? (160 * 1024) / smem_std : 8;
    constexpr int slots = 256 * blocks_per_cu;       // resident workgroups on 256 CUs
    IGemmArgs a = a_in;
    a.par_nb = par_slots(a, BM);
    const int smem = NST * (BM + BN) * 128 + (GLDS ? par_bytes(BN, a.par_nb > PAR_NB ? a.par_nb : PAR_NB) : 0);
    if (smem > 160 * 1024) {
        // feature maps under 8 x 8 with a time embedding (images under 64 px per side at the bottom level): a big tile spans more
        // batches than its LDS can stage rows for - the 64 x 64 tile (<= 66 rows of 64 floats) always fits
        if constexpr (WM * WTM > 64 || WN * WTN > 64 || NST != 2) { g_last_hint_applied = 0; return launch_cfg_amode<2, 2, 32, 32, GLDS, AMODE, 2>(a_in, stream); }
        else { cfgpp_set_error("igemm: %d time-embedding rows per tile do not fit the LDS", a.par_nb); return -2; }
    }
    static int attr_smem = 0;
    auto kern = igemm_kernel<WM, WN, WTM, WTN, GLDS, AMODE, NST>;
    if (smem > attr_smem) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    const int T = cdiv(a.M, BM) * cdiv(a.N, BN);
    const int KT = a.K >> 6;
    a.n_main = T; a.ksplit = 1; a.ws = nullptr; a.staged_epi = g_staged_epi;
    {   // weight bytes vs unique activation bytes (a 3x3 conv re-reads each pixel through L2: its A operand is M x Cin)
        const double w_bytes = 2.0 * a.N * a.K, a_bytes = 2.0 * a.M * (a.C0 + a.C1) * (a.amode == 2 ? 4.0 : a.amode == 3 ? 0.25 : 1.0);
        a.n_major = (g_n_major == 1 || (g_n_major < 0 && w_bytes > 1.5 * a_bytes && cdiv(a.N, BN) >= 8)) ? 1 : 0;
        if (g_n_major < 0 && a.walk_hint) a.n_major = a.walk_hint == 2 ? 1 : 0;
    }
    // K-split: every tile is split S ways into fp32 partials (coalesced, register order) and igemm_reduce_kernel
    // finishes them.  (a) igemm_launch's big-tile rule / diagnostics pass S in IGemmArgs::split; (b) tiny grids with a
    // long K on the 64x64-wave tiles (the 8x8-level convs: 80 tiles on 256 CUs, K = 11520..23040): S is chosen so
    // that T*S fills the resident slots once or twice.
    {
        int S = 0;
        if (a_in.split >= 2 && a.epi == EPI_STORE) S = a_in.split < KT ? a_in.split : KT;
        else if (g_tail_split && a_in.allow_split && (WTM == 64 && WTN == 64) && a.epi == EPI_STORE && T * 2 <= slots && KT >= 32) {
            // at most ONE round of resident workgroups: rounding up (80 tiles x 7 slices = 560 workgroups on 512 slots) left
            // a second round of 48 stragglers as long as the first (g_tail_split == 2 keeps that for A/B)
            S = g_tail_split == 2 ? (slots + T - 1) / T : slots / T;
            if (KT / S < 12) S = KT / 12;
            if (S > 16) S = 16;
        }
        if (S >= 2 && (long)T * S * BM * BN * 4 <= (a_in.ws_buf ? a_in.ws_bytes : WS_BYTES)) {
            float* const ws = a_in.ws_buf ? a_in.ws_buf : ws_for(stream);
            if (ws) { a.n_main = 0; a.ksplit = S; a.ws = ws; }
        }
    }
    const int n_tail = T - a.n_main;
    stats_decide(a, WM * WN * 32 * (WTN * 2 + 16) <= NST * (BM + BN) * 128 && a.staged_epi && n_tail == 0);
    if (n_tail > 0) a.n_major = 0;                     // K-split tiles keep the M-major numbering the reduce kernel uses
    a.walk_div = a.n_major ? cdiv(a.M, BM) : cdiv(a.N, BN);
    walk_plan(a, BM, BN, cdiv(a.M, BM), cdiv(a.N, BN), smem);
    a.tl = tl_take(WM * 100 + WN * 10 + (GLDS ? 1 : 0), a.n_main + n_tail * a.ksplit, NTHR, BM, BN, NST, a);
    hipLaunchKernelGGL(kern, dim3(a.n_main + n_tail * a.ksplit), dim3(NTHR), smem, stream, a);
    if (n_tail > 0)
        hipLaunchKernelGGL((igemm_reduce_kernel<WM, WN, WTM, WTN>), dim3(n_tail, (WTM / 32) * (WTN / 32)), dim3(NTHR), 0, stream, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int WM, int WN, int WTM, int WTN, bool GLDS, int NST = 2>
int launch_cfg(const IGemmArgs& a, hipStream_t stream) {
    switch (a.amode) {
        case 0: return launch_cfg_amode<WM, WN, WTM, WTN, GLDS, 0, NST>(a, stream);
        case 1: return launch_cfg_amode<WM, WN, WTM, WTN, GLDS, 1, NST>(a, stream);
        case 2: return launch_cfg_amode<WM, WN, WTM, WTN, GLDS, 2, NST>(a, stream);
        case 3: return launch_cfg_amode<WM, WN, WTM, WTN, GLDS, 3, NST>(a, stream);
        default: cfgpp_set_error("igemm: bad amode %d", a.amode); return -2;
    }
}

// 32-deep K-tiles (tile32_kernel): any activation map; whole tiles
template <int WM, int WN, int WTM, int WTN, int NST, int WPE, int AMODE>
int launch_tile32_amode(const IGemmArgs& a_in, hipStream_t stream) {
    constexpr int BM = WM * WTM, BN = WN * WTN, NTHR = 64 * WM * WN;
    constexpr int WGS = WPE * 4 / (WM * WN);              // workgroups that are meant to share a CU
    static_assert((NST * (BM + BN) * 64 + par_bytes(BN)) * WGS <= 160 * 1024, "the workgroups that are meant to share a CU do not fit its LDS");
    IGemmArgs a = a_in;
    a.par_nb = par_slots(a, BM);
    const int nb = a.temb ? (a.par_nb > PAR_NB ? a.par_nb : PAR_NB) : 0;
    const int smem = NST * (BM + BN) * 64 + par_bytes(BN, nb);
    if (smem > 160 * 1024) { g_last_hint_applied = 0; return launch_cfg_amode<2, 2, 32, 32, true, AMODE, 2>(a_in, stream); }   // (feature maps under 8 x 8: see launch_cfg_amode; the candidate did not run)
    if (WGS > 1 && smem * WGS > 160 * 1024) {
        // more time-embedding rows than the 2 - 3 workgroups per CU were sized for (small feature maps): the config would run
        // in a regime it was not built for - take the 128 x 128 tile and tell the tuner that the candidate did not run
        g_last_hint_applied = 0;
        return launch_cfg_amode<2, 2, 64, 64, true, AMODE, 2>(a_in, stream);
    }
    static int attr_smem = 0;
    auto kern = tile32_kernel<WM, WN, WTM, WTN, NST, WPE, AMODE>;
    if (smem > attr_smem) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    const int ntm = cdiv(a.M, BM), ntn = cdiv(a.N, BN);
    a.n_main = ntm * ntn; a.ksplit = 1; a.ws = nullptr; a.staged_epi = g_staged_epi;
    const double w_bytes = 2.0 * a.N * a.K, a_bytes = 2.0 * a.M * (a.C0 + a.C1) * (a.amode == 2 ? 4.0 : a.amode == 3 ? 0.25 : 1.0);
    a.n_major = (g_n_major == 1 || (g_n_major < 0 && w_bytes > 1.5 * a_bytes && ntn >= 8)) ? 1 : 0;
    if (g_n_major < 0 && a.walk_hint) a.n_major = a.walk_hint == 2 ? 1 : 0;
    a.walk_div = a.n_major ? ntm : ntn;
    walk_plan(a, BM, BN, ntm, ntn, smem);
    stats_decide(a, NTHR / 64 * 32 * (WTN * 2 + 16) <= NST * (BM + BN) * 64 && a.staged_epi);
    a.tl = tl_take(3200 + WM * 100 + WN * 10, a.n_main, NTHR, BM, BN, NST, a);
    hipLaunchKernelGGL(kern, dim3(a.n_main), dim3(NTHR), smem, stream, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int WM, int WN, int WTM, int WTN, int NST, int WPE>
int launch_tile32(const IGemmArgs& a, hipStream_t stream) {
    switch (a.amode) {
        case 0: return launch_tile32_amode<WM, WN, WTM, WTN, NST, WPE, 0>(a, stream);
        case 1: return launch_tile32_amode<WM, WN, WTM, WTN, NST, WPE, 1>(a, stream);
        case 2: return launch_tile32_amode<WM, WN, WTM, WTN, NST, WPE, 2>(a, stream);
        case 3: return launch_tile32_amode<WM, WN, WTM, WTN, NST, WPE, 3>(a, stream);
        default: cfgpp_set_error("igemm: bad amode %d", a.amode); return -2;
    }
}

}  // namespace

// forced tile config for tests / tuning: 0 = heuristic; 1..8, 10 tile shapes; +20 (21..23) = register-staged
// variant of the same tile (the LDS-DMA variant is the default)
// 128 x 160 tile on 8 waves of 16x16x32 MFMAs (igemm16_kernel): whole tiles, plain-store epilogue
static int g_mf16_heads = 1;           // 1: EPI_HEADS launches may use the tile too (head-major epilogue for the 16 x 16 layout; first run
                                       // on hardware in round 3: Q projection M = 4096 x N = 1280 432 -> 557 TF/s in situ)
extern "C" void cfgpp_igemm_set_mf16_heads(int on) { g_mf16_heads = on ? 1 : 0; }
static bool mf16_supports(const IGemmArgs& a) {
    if (a.N % 160 != 0 || !g_staged_epi || a.K < 64) return false;
    if (a.epi == EPI_STORE) return true;
    return a.epi == EPI_HEADS && g_mf16_heads && a.rows_per_batch % 32 == 0 && a.M % 32 == 0 && a.part_width % 16 == 0 &&
           a.head_dim % 8 == 0;
}
template <int AMODE, int NST>
int launch_mf16_amode(const IGemmArgs& a_in, hipStream_t stream) {
    IGemmArgs a = a_in;
    a.par_nb = par_slots(a, 128);
    const int smem = NST * (128 + 160) * 128 + par_bytes(160, a.par_nb > PAR_NB ? a.par_nb : PAR_NB);
    if (smem > 160 * 1024) { g_last_hint_applied = 0; return launch_cfg_amode<4, 1, 32, 160, true, AMODE, 2>(a_in, stream); }   // (feature maps under 8 x 8: see launch_cfg_amode)
    static int attr_smem = 0;
    auto kern = igemm16_kernel<AMODE, NST>;
    if (smem > attr_smem) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem = smem;
    }
    const int ntm = cdiv(a.M, 128), ntn = a.N / 160;
    a.n_main = ntm * ntn; a.ksplit = 1; a.ws = nullptr; a.staged_epi = 1;
    const double w_bytes = 2.0 * a.N * a.K, a_bytes = 2.0 * a.M * (a.C0 + a.C1) * (a.amode == 2 ? 4.0 : a.amode == 3 ? 0.25 : 1.0);
    a.n_major = (g_n_major == 1 || (g_n_major < 0 && w_bytes > 1.5 * a_bytes && ntn >= 8)) ? 1 : 0;
    if (g_n_major < 0 && a.walk_hint) a.n_major = a.walk_hint == 2 ? 1 : 0;
    a.walk_div = a.n_major ? ntm : ntn;
    walk_plan(a, 128, 160, ntm, ntn, smem);
    stats_decide(a, true);
    a.tl = tl_take(16, a.n_main, 512, 128, 160, NST, a);
    hipLaunchKernelGGL(kern, dim3(a.n_main), dim3(512), smem, stream, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int NST>
int launch_mf16(const IGemmArgs& a, hipStream_t stream) {
    switch (a.amode) {
        case 0: return launch_mf16_amode<0, NST>(a, stream);
        case 1: return launch_mf16_amode<1, NST>(a, stream);
        case 2: return launch_mf16_amode<2, NST>(a, stream);
        case 3: return launch_mf16_amode<3, NST>(a, stream);
        default: cfgpp_set_error("igemm: bad amode %d", a.amode); return -2;
    }
}

// one wave per SIMD (big4_kernel.hip): 256 x 256 / 128 x 320 / 128 x 256 on four waves; whole tiles, staged epilogues
static int big4_smem(int cfg, const IGemmArgs& a) {
    int BM, BN; big4_tile(cfg, &BM, &BN);
    const int nb = a.temb ? par_slots(a, BM) : 0;
    return 2 * (BM + BN) * 128 + big4_par_bytes(BN, nb);
}
static bool big4_ok(int cfg, const IGemmArgs& a) { return big4_supports(cfg, a, g_staged_epi != 0) && big4_smem(cfg, a) <= 160 * 1024; }
static int launch_big4(int cfg, const IGemmArgs& a_in, hipStream_t stream) {
    int BM, BN; big4_tile(cfg, &BM, &BN);
    IGemmArgs a = a_in;
    a.par_nb = par_slots(a, BM);
    const int smem = big4_smem(cfg, a);
    const int ntm = cdiv(a.M, BM), ntn = cdiv(a.N, BN);
    a.n_main = ntm * ntn; a.ksplit = 1; a.ws = nullptr; a.staged_epi = 1;
    const double w_bytes = 2.0 * a.N * a.K, a_bytes = 2.0 * a.M * (a.C0 + a.C1) * (a.amode == 2 ? 4.0 : a.amode == 3 ? 0.25 : 1.0);
    a.n_major = (g_n_major == 1 || (g_n_major < 0 && w_bytes > 1.5 * a_bytes && ntn >= 8)) ? 1 : 0;
    if (g_n_major < 0 && a.walk_hint) a.n_major = a.walk_hint == 2 ? 1 : 0;
    a.walk_div = a.n_major ? ntm : ntn;
    walk_plan(a, BM, BN, ntm, ntn, smem);
    stats_decide(a, true);
    a.tl = tl_take(4000 + cfg, a.n_main, 256, BM, BN, 2, a);
    return big4_run(cfg, a, a.n_main, smem, stream);
}

static bool big4p_ok(const IGemmArgs& a) { return big4p_supports(a, g_staged_epi != 0) && big4p_smem(big4_par_bytes(256, 0)) <= 160 * 1024; }
static int launch_big4p(const IGemmArgs& a_in, hipStream_t stream) {
    IGemmArgs a = a_in;
    a.par_nb = 0;                                      // (no time-embedding rows on a token-major linear: big4p_supports)
    const int pb = big4_par_bytes(256, 0), smem = big4p_smem(pb);
    const int ntm = cdiv(a.M, 256), ntn = cdiv(a.N, 256);
    a.n_main = ntm * ntn; a.ksplit = 1; a.ws = nullptr; a.staged_epi = 1; a.tl = nullptr;
    const double w_bytes = 2.0 * a.N * a.K, a_bytes = 2.0 * a.M * a.C0;
    a.n_major = (g_n_major == 1 || (g_n_major < 0 && w_bytes > 1.5 * a_bytes && ntn >= 8)) ? 1 : 0;
    if (g_n_major < 0 && a.walk_hint) a.n_major = a.walk_hint == 2 ? 1 : 0;
    a.walk_div = a.n_major ? ntm : ntn;
    walk_plan(a, 256, 256, ntm, ntn, smem);
    stats_decide(a, true);
    return big4p_run(a, a.n_main < 256 ? a.n_main : 256, smem, pb, stream);
}

static int g_force_cfg = 0;
static int g_staging = 1;          // 1 = global_load_lds (default), 0 = register staging
extern "C" void cfgpp_igemm_force_config(int cfg) { g_force_cfg = cfg; }
extern "C" void cfgpp_igemm_set_staging(int glds) { g_staging = glds ? 1 : 0; }
extern "C" void cfgpp_igemm_set_tail_split(int on) { g_tail_split = on == 2 ? 2 : on ? 1 : 0; }

static int launch_config(int cfg, const IGemmArgs& a, hipStream_t stream) {
    bool glds = g_staging != 0;
    if (cfg > 20 && cfg < 24) { cfg -= 20; glds = false; }
    switch (cfg) {
        case 1: return glds ? launch_cfg<2, 2, 64, 64, true>(a, stream) : launch_cfg<2, 2, 64, 64, false>(a, stream);
        case 2: return glds ? launch_cfg<4, 1, 64, 64, true>(a, stream) : launch_cfg<4, 1, 64, 64, false>(a, stream);
        case 3: return glds ? launch_cfg<2, 2, 32, 32, true>(a, stream) : launch_cfg<2, 2, 32, 32, false>(a, stream);
        // 8-wave, one-workgroup-per-CU tiles: fewer tile bytes per FLOP through the CU's memory->LDS path
        case 4: return launch_cfg<2, 4, 128, 64, true>(a, stream);     // 256 x 256
        case 5: return launch_cfg<4, 2, 64, 160, true>(a, stream);     // 256 x 320 (every SD/SDXL channel count is k*320)
        case 6: return launch_cfg<4, 2, 64, 64, true>(a, stream);      // 256 x 128
        // N = k*160 tiles for the mid-size levels, where 256-wide tiles leave CUs idle (M*N / 256 CUs = 128 x 160
        // at M = 4096, N = 1280 and 128 x 320 at M = 16384, N = 640)
        case 7: return launch_cfg<4, 1, 32, 160, true>(a, stream);     // 128 x 160, 4 waves, 2 workgroups / CU
        case 8: return launch_cfg<4, 2, 32, 160, true>(a, stream);     // 128 x 320, 8 waves
        // Deeper LDS rings of the small tiles (3 / 4 stages, counted vmcnt + raw s_barrier): with the operands hot in L2 they
        // are +-1 % of the 2-stage form (profiles/r02/ab/igemm_ring_variants_run5.txt), but inside a forward every launch
        // streams its weights from HBM / the Infinity Cache, where one tile of lookahead (640 MFMA cycles on a 4-wave tile)
        // is shorter than the miss latency.  Only the in-situ tuner picks them (same K order: bit-identical results).
        // (two K-tiles per barrier on a 4-stage ring was 9 .. 30 % slower and stays dropped)
        case 9: return launch_cfg<4, 1, 32, 160, true, 3>(a, stream);  // 128 x 160, 3 stages (110 KB, 1 workgroup / CU)
        case 11: return launch_cfg<4, 1, 32, 160, true, 4>(a, stream); // 128 x 160, 4 stages (147 KB)
        case 12: return launch_cfg<2, 2, 64, 64, true, 3>(a, stream);  // 128 x 128, 3 stages (96 KB)
        case 14: return launch_cfg<4, 2, 64, 64, true, 3>(a, stream);  // 256 x 128, 3 stages (144 KB)
        // (measured and not kept as tuner candidates, profiles/r02/ab/igemm_insitu_run8.txt: 128 x 256 on 3 stages, 128 x 128 as
        //  8 waves of 32 x 64 with two workgroups per CU, 256 x 64 - each pinned for 0 .. 2 launches of a forward)
        // 256 x 320 with the 8 waves stacked along M (32 x 320 per wave, 10 accumulator tiles): a wave holds whole
        // (value | gate) column pairs, so the GEGLU projections (N = 8C = k * 320) can use the 320-wide tile too.
        // Pinned by the tuner for GEGLU launches only (a plain store would not fit its LDS-staged epilogue).
        case 10: return launch_cfg<8, 1, 32, 320, true>(a, stream);
        // 32-deep K-tiles (tile32_kernel).  15 / 16 / 17: three stages, 2 - 3 workgroups per CU; 13 / 20: the 256-wide tiles on FOUR
        // stages.  Tuner candidates (same k order as every 32x32x16 tile).  (Numbers above 20 mean "register-staged variant of
        // config - 20", see above.)
        case 15: return launch_tile32<4, 2, 64, 64, 3, 4>(a, stream);      // 256 x 128, 8 waves, 72 KB: 2 / CU
        case 16: return launch_tile32<2, 2, 64, 64, 3, 3>(a, stream);      // 128 x 128, 4 waves, 48 KB: 3 / CU
        case 17: return launch_tile32<4, 1, 64, 64, 3, 2>(a, stream);      // 256 x 64, 4 waves, 60 KB: 2 / CU
        case 13: return launch_tile32<2, 4, 128, 64, 4, 2>(a, stream);     // 256 x 256, 8 waves, 128 KB
        // 128 x 320 as FOUR waves of 64 x 160 on a two-stage 32-deep ring (65 KB): TWO workgroups per CU, i.e. the 256 x 320 tile of
        // config 5 (same wave tiles, same LDS traffic per CU) cut into two halves that no longer share a barrier - while one half
        // waits for its tile or sits in its epilogue, the other half's waves have the matrix pipes (round 5).  Plain stores only
        // (ten accumulator tiles per wave: no staged head-major epilogue, no GEGLU pairs).
        case 27: if (a.epi != EPI_STORE) return launch_tile32<2, 2, 64, 64, 3, 3>(a, stream);
                 return launch_tile32<2, 2, 64, 160, 2, 2>(a, stream);
        case 20: if (a.epi == EPI_GEGLU) return launch_tile32<2, 4, 128, 64, 4, 2>(a, stream);       // (GEGLU needs 64-wide wave tiles)
                 return launch_tile32<4, 2, 64, 160, 4, 2>(a, stream);     // 256 x 320, 8 waves, 147 KB
        // 128 x 160 as 8 waves of 32 x 80 on the 16x16x32 MFMA, 3 / 4 LDS stages (igemm16_kernel): plain-store launches with
        // N % 160 == 0 only - anything else falls back to the 4-wave 128 x 160 tile.  Not a tuner candidate (different k order).
        case 18: return mf16_supports(a) ? launch_mf16<3>(a, stream) : launch_cfg<4, 1, 32, 160, true>(a, stream);
        case 19: return mf16_supports(a) ? launch_mf16<4>(a, stream) : launch_cfg<4, 1, 32, 160, true>(a, stream);
        // one wave per SIMD (big4_kernel.hip); launches they do not admit (generic epilogues, > 4 GiB operands) take the 128 x 128 tile
        case 24: case 25: case 26: return big4_ok(cfg, a) ? launch_big4(cfg, a, stream) : launch_cfg<2, 2, 64, 64, true>(a, stream);
        // the 256 x 256 one-wave-per-SIMD tile as a persistent kernel with the next output tile's first K-tile prefetched under the
        // epilogue (big4p_kernel.hip): token-major linears only
        case 28: return big4p_ok(a) ? launch_big4p(a, stream) : launch_cfg<2, 2, 64, 64, true>(a, stream);
        default: cfgpp_set_error("igemm: bad config %d", cfg); return -2;
    }
}

// ---- in-situ tile tuning -------------------------------------------------------------------------
// The engines time every igemm launch of their plan in place (HIP events between the launches of a real
// forward) once per candidate tile config and pin the fastest per launch through IGemmArgs::cfg_hint
// (engine_base.h, tune_plan).  Every non-split config runs the K loop in the same order, so the RESULT does
// not depend on the choice (bit-identical); K-split launches (different summation order) stay rule-based.
static int g_autotune = 1;
extern "C" void cfgpp_igemm_set_autotune(int on) { g_autotune = on ? 1 : 0; }
// tile of the rule-based K-split launches: 14 (256 x 128 on 3 stages; 8x8-level convs 525 -> 619 TF/s in situ against the
// 2-stage 128 x 128 tile, profiles/r02/ab/igemm_insitu_run8.txt), 1 (128 x 128, 2 stages) or 12 (128 x 128, 3 stages)
static int g_split_cfg = 14;
extern "C" void cfgpp_igemm_set_split_tile(int cfg) { g_split_cfg = (cfg == 1 || cfg == 12) ? cfg : 14; }
static int g_mf16 = 4;                 // 0 = off; 3 / 4 = the 8-wave 16x16x32-MFMA 128 x 160 tile (3 / 4 stages) by rule
extern "C" void cfgpp_igemm_set_mf16(int mode) { g_mf16 = (mode == 3 || mode == 4) ? mode : 0; }
static int g_mf16_linear = 1;          // 1: the rule also takes token-major linears (amode 0); 0: convolutions only, the tuner picks the linears' tile (A/B)
extern "C" void cfgpp_igemm_set_mf16_linear(int on) { g_mf16_linear = on ? 1 : 0; }
static int g_mf16_rounds = 2;          // the rule also takes grids of exactly 2 .. n full rounds of 256 tiles (1 = one round only).  Two rounds
                                       // = the M = 16384 x N = 640 class (32x32 level of SD1.5 at batch 8, 64x64 level of SDXL at batch 2): convs
                                       // 743 -> 860, 810 -> 954 TF/s in situ with the round-3 K-tile schedule (profiles/r03/ab/mf16_rounds.txt);
                                       // three rounds (QKV projection N = 3840 at M = 4096) lose against the 256 x 256 tile
extern "C" void cfgpp_igemm_set_mf16_rounds(int n) { g_mf16_rounds = n >= 1 ? n : 1; }
static int g_force_split = 0;          // diagnostics: with a forced config, K-split every tile this many ways
extern "C" void cfgpp_igemm_force_split(int s) { g_force_split = s >= 2 ? s : 0; }
// big-tile K-split rule: least K-tiles per slice (0 = rule off, the default: inside a forward the 3-stage 256 x 128 tile the
// tuner pins for these launches beat it, 751 vs 656 TF/s on the 16x16-level convs - profiles/r02/ab/igemm_insitu_run6.txt -
// although the split wins 8 .. 21 % when the same launch is timed alone with hot caches)
static int g_big_split_min_kt = 0;
extern "C" void cfgpp_igemm_set_big_split(int min_kt) { g_big_split_min_kt = min_kt > 0 ? min_kt : 0; }
int igemm_autotune_enabled() { return g_autotune && g_force_cfg == 0 && g_staging != 0; }
// bit c: the tuner may pin tile config c; bit 31: the tile-walk stage runs.  Default: everything but 25 / 26 (big4_kernel.hip: 128 x 320 / 128 x 256 on four waves) and 27 (128 x 320 as two workgroups per CU) - measured
// in situ on the MI355X (profiles/r05/ab/): forced per launch they are slower than the tuned plan on every launch of the SD1.5 /
// SDXL forwards at the bench batches, and offered to the tuner they are never pinned (the 256 x 256 K loop does reach 0.75 - 0.80 of
// the matrix peak, but no launch with N = k * 320 fits the tile, and four waves take twice as long over the epilogue as eight), so
// they would only lengthen the tuning passes (27: 11 - 14 % slower than the 8-wave 256 x 320 tile on the 64x64-level convolutions it
// was built for, profiles/r05/ab/forced_hint27_*); the 256 x 256 one (24) stays a candidate because the VAE decoder's N = 256 / 512 convolutions
// do take it (+8 % on those launches, decode -2.5 %: profiles/r05/ab/vae_big4_call7.txt).  cfgpp_igemm_set_tune_mask(0xffffffff) offers all.
static unsigned g_tune_mask = 0xf1ffffffu;
extern "C" void cfgpp_igemm_set_tune_mask(unsigned mask) { g_tune_mask = mask; }
unsigned igemm_tune_mask() { return g_tune_mask; }
// every switch that changes what the tuner measures or may pin, folded into one word: the Python pin cache persists pins only
// while this still has the value it had when the engine was built (cfgpp_amd/tune_cache.py)
extern "C" unsigned cfgpp_igemm_tuner_state(void) {
    return g_tune_mask ^ ((unsigned)g_big_tiles << 1) ^ ((unsigned)g_tail_split << 3) ^ ((unsigned)(g_n_major + 1) << 6) ^ ((unsigned)g_walk_blocked << 9) ^
           ((unsigned)g_staged_epi << 10) ^ ((unsigned)g_force_cfg << 12) ^ ((unsigned)g_staging << 20);
}

// Arms the timeline: the `target`-th igemm_launch call from now on (0-based) records 16 x uint64 per workgroup (slots: see
// IGemmArgs::tl) into buf[grid][16] (device memory, >= cap_blocks * 128 bytes; launches with more workgroups than cap_blocks
// are not recorded); buf = null disarms.
extern "C" void cfgpp_igemm_timeline(void* buf, long cap_blocks, int target) {
    g_tl = (unsigned long long*)buf; g_tl_cap = buf ? cap_blocks : 0; g_tl_target = buf ? target : -1; g_tl_count = 0;
}
extern "C" void cfgpp_igemm_timeline_info(int* out12) { for (int i = 0; i < 12; ++i) out12[i] = g_tl_info[i]; }

// 1 when the last igemm_launch ran the tile its cfg_hint asked for (or had no hint); 0 when the hint was not applicable to that
// launch (rule-based K-split / 16x16x32 tile, GEGLU on a 320-wide tile, ...) and the heuristic tile ran instead: the tuner must
// not record a time for a candidate that did not run.
int igemm_last_hint_applied() { return g_last_hint_applied; }

int igemm_launch(const IGemmArgs& a_in, hipStream_t stream) {
    IGemmArgs a = a_in;
    g_last_hint_applied = (a_in.cfg_hint & 63) == 0 ? 1 : 0;
    a.allow_split = 1;
    a.tl = nullptr;
    if (g_tl) { if (g_tl_count == g_tl_target) a.tl = g_tl; ++g_tl_count; }
    const int Cin = a.C0 + a.C1;
    CFGPP_REQUIRE(a.C0 > 0 && a.C0 % 64 == 0 && a.C1 % 64 == 0, "igemm: C0=%d C1=%d must be multiples of 64", a.C0, a.C1);
    CFGPP_REQUIRE(a.taps == 1 || a.taps == 9, "igemm: taps=%d", a.taps);
    CFGPP_REQUIRE(a.K == a.taps * Cin, "igemm: K=%d != taps*Cin=%d", a.K, a.taps * Cin);
    CFGPP_REQUIRE(a.N % 4 == 0 && a.M > 0 && a.N > 0, "igemm: M=%d N=%d (N must be a multiple of 4)", a.M, a.N);
    CFGPP_REQUIRE(a.amode == 0 || (a.H > 0 && a.W > 0 && a.rows_per_batch == a.H * a.W), "igemm: spatial args");
    CFGPP_REQUIRE(a.amode != 3 || (a.H < 2048 && a.W < 2048 && (a.M / a.rows_per_batch) < 512), "igemm: upsample range");
    CFGPP_REQUIRE(a.epi != EPI_GEGLU || a.N % 64 == 0, "igemm: GEGLU needs N %% 64 == 0");
    CFGPP_REQUIRE(a.epi != EPI_HEADS || (a.head_dim % 4 == 0 && a.part_width % 4 == 0), "igemm: heads args");
    CFGPP_REQUIRE(a.rows_per_batch <= 0 || a.M / a.rows_per_batch < (1 << 20), "igemm: %d rows in batches of %d (batch index must stay below 2^20)", a.M, a.rows_per_batch);
    // tile heuristic: 128x128 (2x2 waves of 64x64) when it fills the chip, 256x64 for
    // N = 64*odd (e.g. 320), 64x64 (4 waves of 32x32) for small problems.
    int cfg = g_force_cfg;
    if (cfg == 0) {
        const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
        const long t256x64 = (long)cdiv(a.M, 256) * cdiv(a.N, 64);
        const bool n_odd64 = (a.N % 128) != 0;
        const int KT = a.K >> 6;
        // 8-wave one-block-per-CU tiles (fewer tile bytes per FLOP): only when they fill the 256 CUs
        const long t5 = (a.N % 320 == 0) ? (long)cdiv(a.M, 256) * (a.N / 320) : 0;
        const long t4 = (a.N % 256 == 0) ? (long)cdiv(a.M, 256) * (a.N / 256) : 0;
        auto fills = [](long t) { return t >= 224 && (t % 256 == 0 || t % 256 >= 160 || t >= 1024); };
        if (g_big_tiles && a.epi != EPI_GEGLU && a.K >= 1280 && fills(t5)) cfg = 5;
        else if (g_big_tiles && t4 >= 192 && (t4 % 256 == 0 || t4 % 256 >= 128 || t4 >= 512)) cfg = 4;
        else
        if (t128 >= 256 && !n_odd64) cfg = 1;
        else if (t256x64 >= 256 && (n_odd64 || a.N <= 64)) cfg = 2;
        else if (t128 >= 200) cfg = 1;
        else if (g_tail_split && KT >= 32 && t128 >= 8 && a.epi == EPI_STORE) cfg = g_split_cfg;   // few tiles, long K: K-split them
        else cfg = 3;
        if (a.epi == EPI_GEGLU && cfg == 3) cfg = 1;   // GEGLU needs 64-wide wave tiles
    }
    // Big-tile K-split (rule-based like the 8x8-level split, so the result never depends on tuning): M x N is too small
    // for the 8-wave tiles to fill 256 CUs (M = 4096, N = 1280: 128 tiles of 128 x 320, and the 4-wave 128 x 160 tile
    // that does fill them runs one wave per SIMD at 600-650 TF/s) but K is long (the 16x16-level convs, K = 5760 .. 23040,
    // and the feed-forward output projection, K = 5120): 128 x 320 tiles, each split 256 / T ways.
    bool big_split = false;
    a.split = 0;
    if (g_force_cfg != 0) a.split = g_force_split;
    else if (g_big_split_min_kt > 0 && g_big_tiles && g_staging != 0 && g_tail_split && a.epi == EPI_STORE && a.N % 320 == 0) {
        const long t8 = (long)cdiv(a.M, 128) * (a.N / 320);
        const int KT = a.K >> 6;
        if (t8 >= 64 && t8 <= 128) {
            int S = (int)(256 / t8);
            while (S >= 2 && KT / S < g_big_split_min_kt) --S;
            if (S >= 2 && t8 * S >= 192) { cfg = 8; a.split = S; big_split = true; }
        }
    }
    // 16x16x32-MFMA tile by rule (never by tuning: it sums k in a different order): plain-store launches whose 128 x 160
    // grid is ONE round of 200 .. 256 tiles - the M = 4096 x N = 1280 class (16x16 level of SD1.5 at batch 8, 32x32 level of
    // SDXL at batch 2).  In situ against the tuned 3-stage 256 x 128 tile: convs 762 -> 866 TF/s, FF-out K = 5120 693 -> 800,
    // to_out K = 1280 436 -> 505, SD1.5 forward 21.06 -> 20.61 ms (profiles/r02/ab/igemm_mf16_run9.txt).
    if (g_force_cfg == 0 && g_mf16 != 0 && g_staging != 0 && !big_split && mf16_supports(a) && (g_mf16_linear || a.amode != 0)) {
        const long t7 = (long)cdiv(a.M, 128) * (a.N / 160);
        // (also grids of exactly 2 .. g_mf16_rounds full rounds; default 2, see g_mf16_rounds)
        const bool full_rounds = t7 > 256 && t7 % 256 == 0 && t7 / 256 <= g_mf16_rounds;
        if ((t7 >= 200 && t7 <= 256) || full_rounds) return launch_config(g_mf16 == 4 ? 19 : 18, a, stream);
    }
    if (g_force_cfg == 0 && a.cfg_hint > 0 && g_staging != 0 && !big_split) {
        const int KT = a.K >> 6;
        const long t128 = (long)cdiv(a.M, 128) * cdiv(a.N, 128);
        // (must say "splits" exactly when launch_cfg_amode will: T * 2 <= resident slots of that tile)
        const long t_rule = cfg == 14 ? (long)cdiv(a.M, 256) * cdiv(a.N, 128) : t128;
        const bool rule_splits = (cfg == 1 || cfg == 12 || cfg == 14) && g_tail_split && a.epi == EPI_STORE && KT >= 32 && t_rule * 2 <= (cfg == 1 ? 512 : 256);
        const int h = a.cfg_hint & 63;
        const bool valid = (h == 1 || h == 4 || h == 6 || h == 12 || h == 14 || (h == 10 && a.epi == EPI_GEGLU) ||
                            ((h == 5 || h == 7 || h == 8 || h == 9 || h == 11) && a.epi != EPI_GEGLU) ||
                            h == 13 || h == 15 || h == 16 || h == 17 || (h == 20 && a.epi != EPI_GEGLU) || (h == 27 && a.epi == EPI_STORE) || ((h >= 24 && h <= 26) && big4_ok(h, a)) || (h == 28 && big4p_ok(a))) && (g_big_tiles || h == 1);
        if (!rule_splits && valid) { cfg = h; a.allow_split = 0; g_last_hint_applied = 1; }
    }
    a.walk_hint = (g_force_cfg == 0 && g_staging != 0) ? (a.cfg_hint >> 6) & 3 : 0;      // tuner-pinned tile walk (0 = by operand bytes)
    return launch_config(cfg, a, stream);
}
