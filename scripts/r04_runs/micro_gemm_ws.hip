// Stand-alone K loops of an LDS-DMA GEMM tile: does a loader-wave / MFMA-wave split lift it?  An early barrier?  Bigger wave tiles?  (a 1 KB LDS-DMA piece costs the
// issuing wave 60-185 cycles, and in the shipped kernels the waves that own accumulators issue them.)
// One 256 x 128 fp16 tile per workgroup, 64-deep K-tiles, 3-stage LDS ring (48 KB per stage: A rows [256][128 B] then B rows
// [128][128 B], XOR-swizzled 16-byte chunks, 8-row DMA pieces - the layouts of cfgpp_amd/csrc), 8 MFMA waves of 64 x 64 each
// (16 x v_mfma_f32_32x32x16_f16 per K-tile and wave, fragments one 16-deep step ahead), one barrier per K-tile, counted vmcnt.
//   NLOAD = 0: every MFMA wave also issues its 6 of the 48 pieces per K-tile, spread between its MFMAs (the shipped scheme);
//   NLOAD = 4: four extra waves (one per SIMD) issue all 48 pieces (12 each) and do nothing else; PRIO raises their priority.
// Every workgroup reads the SAME A and B rows (3 MB at K = 4096: L2-resident), so what is measured is the CU-side loop, not the
// memory system.  One workgroup per CU (144 KB of LDS).  Checks block 0's tile against a host reference at a small K first.
// Timing-only rows: the B pieces (a third of the bytes) not loaded; nothing loaded (LDS fragment reads + MFMAs + barrier only).
//   hipcc --offload-arch=gfx950 -O3 -o micro_gemm_ws.bin micro_gemm_ws.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 256, BN = 128, BK = 64, NST = 3, STAGE = (BM + BN) * 128, NPIECE = (BM + BN) / 8;

template <int NLOAD, int PRIO, int SKIP = 0, int SKEW = 0>     // SKIP (timing only, wrong results): 1 = the B pieces are not loaded, 2 = nothing is loaded
__global__ void __launch_bounds__((8 + NLOAD) * 64) gemm_tile(const half_t* __restrict__ A, const half_t* __restrict__ B, float* __restrict__ C,
                                                              unsigned long long* cyc, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = NLOAD > 0 && wid >= 8;
    constexpr int NISSUE = NLOAD > 0 ? NLOAD : 8;                 // waves that issue pieces
    constexpr int PPW = NPIECE / NISSUE;                          // pieces per issuing wave and K-tile (6 or 12)
    const int iw = NLOAD > 0 ? wid - 8 : wid;                      // index among the issuing waves
    const int nkt = K / BK;
    // DMA source offsets (elements) of this wave's pieces: piece p = rows [8p, 8p+8) of the A-then-B row image
    const int r8 = lane >> 3, pc = lane & 7;
    unsigned src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = iw + i * NISSUE, row = p * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        src[i] = row < BM ? (unsigned)(row * K + lc * 8) : (unsigned)((BM * K) + (row - BM) * K + lc * 8);   // B follows A in one allocation
    }
    auto issue_piece = [&](int kt, int i) {
        const int p = iw + i * NISSUE;
        if (SKIP == 2 || (SKIP == 1 && p >= BM / 8)) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)kt * BK + src[i]),
                                         (__attribute__((address_space(3))) void*)(smem + (kt % NST) * STAGE + p * 1024), 16, 0, 0);
    };
    const bool issuer = NLOAD > 0 ? loader : true;
    if (issuer) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(0, i);
        if (nkt > 1) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(1, i);
        }
    }
    if (loader) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nkt) {
#pragma unroll
                for (int i = 0; i < PPW; ++i) issue_piece(kt + 2, i);
            }
        }
        return;
    }
    // ---- MFMA waves ----
    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = (wm * 64 + l31) * 128, brow = (BM + wn * 64 + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int half32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + half32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (SKEW) {
        // SKEW (NLOAD = 0 only): the barrier that publishes tile kt+1 sits BEFORE the last 16-deep step of tile kt, whose fragments are
        // already in registers; right after it the first fragments of tile kt+1 are read, and the four MFMAs of that last step cover
        // their latency - the wave never waits for LDS with an empty matrix pipe.  Pieces of tile kt+2 go out after MFMAs 1,3,5,7,9,11.
        static_assert(NLOAD == 0, "");
        half8_t af[2][2], bf[2][2];
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) { af[0][h] = frag(smem, arow, h, 0); bf[0][h] = frag(smem, brow, h, 0); }
        for (int kt = 0; kt < nkt; ++kt) {
            const char* st = smem + (kt % NST) * STAGE;
            const char* stn = smem + ((kt + 1) % NST) * STAGE;
            const bool more = kt + 2 < nkt;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) { af[(ks + 1) & 1][h] = frag(st, arow, h, ks + 1); bf[(ks + 1) & 1][h] = frag(st, brow, h, ks + 1); }
                } else if (kt + 1 < nkt) {
                    // every LDS read of tile kt has been issued one step ago; lgkmcnt(0) makes sure they also landed before the stage is released
                    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int h = 0; h < 2; ++h) { af[0][h] = frag(stn, arow, h, 0); bf[0][h] = frag(stn, brow, h, 0); }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                        const int m = ks * 4 + i * 2 + j;
                        if (m < 12 && (m & 1) && more) issue_piece(kt + 2, m >> 1);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int kt = 0; kt < nkt; ++kt) {
        if (NLOAD == 0) { if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (kt % NST) * STAGE;
        const bool more = NLOAD == 0 && kt + 2 < nkt;
        half8_t af[2][2], bf[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) { af[0][h] = frag(st, arow, h, 0); bf[0][h] = frag(st, brow, h, 0); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int h = 0; h < 2; ++h) { af[(ks + 1) & 1][h] = frag(st, arow, h, ks + 1); bf[(ks + 1) & 1][h] = frag(st, brow, h, ks + 1); }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    if constexpr (NLOAD == 0) {                    // this wave's 6 pieces of tile kt+2, one after MFMAs 1, 3, 6, 9, 11, 14
                        constexpr int at[6] = {1, 3, 6, 9, 11, 14};
                        const int m = ks * 4 + i * 2 + j;
#pragma unroll
                        for (int q = 0; q < 6; ++q) if (m == at[q] && more) issue_piece(kt + 2, q);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wid == 0) cyc[blockIdx.x] = t1 - t0;
    // C[block][256][128] fp32 (only block 0 is checked; every block stores so that nothing is optimised away)
    float* Cb = C + (long)(blockIdx.x & 1) * BM * BN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // A operand rows = tile rows (lane l31), B operand rows = tile columns: acc[i][j][r] is C[row = 8*(r/4)*... ] in MFMA layout:
                // 32x32 accumulator: column (B index) = lane & 31, row (A index) = (r & 3) + 8 * (r >> 2) + 4 * hi
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = wn * 64 + j * 32 + l31;
                Cb[row * BN + col] = acc[i][j][r];
            }
}

// ---- 256 x 256 tile, FOUR waves of 128 x 128 (one per SIMD, 256 accumulator registers each): 0.5 fragment reads and 0.25 DMA
// pieces per MFMA instead of 1 and 0.375; 64-deep K-tiles on a 2-stage ring (2 x 64 KB); every wave issues 16 pieces per K-tile,
// one after every fourth MFMA.  Same layouts, same check.
constexpr int BIG = 256, BSTAGE = 2 * BIG * 128;
template <bool SKIP>
__global__ void __launch_bounds__(256) gemm_tile_big(const half_t* __restrict__ A, float* __restrict__ C, unsigned long long* cyc, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkt = K / BK;
    const int r8 = lane >> 3, pc = lane & 7;
    unsigned src[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (wid + i * 4) * 8 + r8;
        src[i] = (unsigned)(row * K + ((pc ^ ((row >> 1) & 7)) * 8));
    }
    auto issue_piece = [&](int kt, int i) {
        if (SKIP) return;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (long)kt * BK + src[i]),
                                         (__attribute__((address_space(3))) void*)(smem + (kt & 1) * BSTAGE + (wid + i * 4) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) issue_piece(0, i);
    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = (wm * 128 + l31) * 128, brow = (BIG + wn * 128 + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int q32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + q32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (kt & 1) * BSTAGE;
        const bool more = kt + 1 < nkt;
        half8_t af[2][4], bf[2][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { af[0][h] = frag(st, arow, h, 0); bf[0][h] = frag(st, brow, h, 0); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int h = 0; h < 4; ++h) { af[(ks + 1) & 1][h] = frag(st, arow, h, ks + 1); bf[(ks + 1) & 1][h] = frag(st, brow, h, ks + 1); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    const int m = ks * 16 + i * 4 + j;
                    if ((m & 3) == 3 && more) issue_piece(kt + 1, m >> 2);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wid == 0) cyc[blockIdx.x] = t1 - t0;
    float* Cb = C + (long)(blockIdx.x & 1) * BIG * BIG;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = wn * 128 + j * 32 + l31;
                Cb[row * BIG + col] = acc[i][j][r];
            }
}

// ---- the same 256 x 256 four-wave loop with the piece issue slimmed down: the last K-tile is peeled (no branch around a piece), a
// piece is `s_mov m0 ; global_load_lds_dwordx4 v_off32, s[base]` (scalar tile base + 32-bit per-lane byte offset: no 64-bit vector
// add per piece, and the offset registers are the piece's own), and the 16 pieces go out after every third of the first 48 MFMAs.
template <int DUMMY>
__global__ void __launch_bounds__(256) gemm_tile_big2(const half_t* __restrict__ A, float* __restrict__ C, unsigned long long* cyc, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkt = K / BK;
    const int r8 = lane >> 3, pc = lane & 7;
    unsigned src[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (wid + i * 4) * 8 + r8;
        src[i] = (unsigned)(row * K + ((pc ^ ((row >> 1) & 7)) * 8)) * 2u;       // bytes
    }
    const unsigned lds0 = (unsigned)(size_t)smem + wid * 1024;
    auto issue_piece = [&](int kt, int i) {
        const char* base = reinterpret_cast<const char*>(A) + (long)kt * BK * 2;            // wave-uniform
        const unsigned m0v = lds0 + (kt & 1) * BSTAGE + i * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(src[i]), "s"(base) : "memory");
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) issue_piece(0, i);
    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = (wm * 128 + l31) * 128, brow = (BIG + wn * 128 + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int q32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + q32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };
    auto tile = [&](auto more_c, int kt) {
        constexpr bool MORE = decltype(more_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* st = smem + (kt & 1) * BSTAGE;
        half8_t af[2][4], bf[2][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { af[0][h] = frag(st, arow, h, 0); bf[0][h] = frag(st, brow, h, 0); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int h = 0; h < 4; ++h) { af[(ks + 1) & 1][h] = frag(st, arow, h, ks + 1); bf[(ks + 1) & 1][h] = frag(st, brow, h, ks + 1); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    const int m = ks * 16 + i * 4 + j;
                    if constexpr (MORE) { if (m < 48 && m % 3 == 2) issue_piece(kt + 1, m / 3); }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt + 1 < nkt; ++kt) tile(std::integral_constant<bool, true>{}, kt);
    tile(std::integral_constant<bool, false>{}, nkt - 1);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wid == 0) cyc[blockIdx.x] = t1 - t0;
    float* Cb = C + (long)(blockIdx.x & 1) * BIG * BIG;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = wn * 128 + j * 32 + l31;
                Cb[row * BIG + col] = acc[i][j][r];
            }
}

// ---- ... and with the barrier that publishes tile kt+1 moved BEFORE the last 16-deep step of tile kt (its fragments are already in
// registers): right after it the first fragments of tile kt+1 are read, and that step's 16 MFMAs cover their latency; the 16 pieces
// of tile kt+2 go out under those MFMAs (8) and under the first step of tile kt+1 (8), i.e. as soon as tile kt's stage is free.
template <bool LOAD>
__global__ void __launch_bounds__(256) gemm_tile_big3(const half_t* __restrict__ A, float* __restrict__ C, unsigned long long* cyc, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkt = K / BK;                      // >= 3
    const int r8 = lane >> 3, pc = lane & 7;
    unsigned src[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (wid + i * 4) * 8 + r8;
        src[i] = (unsigned)(row * K + ((pc ^ ((row >> 1) & 7)) * 8)) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)smem + wid * 1024;
    auto issue_piece = [&](int kt, int i) {
        if constexpr (!LOAD) return;
        const char* base = reinterpret_cast<const char*>(A) + (long)kt * BK * 2;
        const unsigned m0v = lds0 + (kt & 1) * BSTAGE + i * 4096;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(src[i]), "s"(base) : "memory");
    };
    const int l31 = lane & 31, hi = lane >> 5, fsw = (l31 >> 1) & 7;
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = (wm * 128 + l31) * 128, brow = (BIG + wn * 128 + l31) * 128;
    auto frag = [&](const char* st, int rowoff, int q32, int ks) {
        return *reinterpret_cast<const half8_t*>(st + rowoff + q32 * 32 * 128 + ((((ks << 1) | hi) ^ fsw) << 4));
    };
    half8_t af[2][4], bf[2][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) issue_piece(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int h = 0; h < 4; ++h) { af[0][h] = frag(smem, arow, h, 0); bf[0][h] = frag(smem, brow, h, 0); }
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(1, i);
    // P0: pieces 8..15 of tile kt+1 under step 0; BAR: tile kt+1 exists (barrier + its first fragments before step 3); P3: pieces 0..7 of tile kt+2 under step 3
    auto tile = [&](auto p0_c, auto bar_c, auto p3_c, int kt) {
        constexpr bool P0 = decltype(p0_c)::value, BAR = decltype(bar_c)::value, P3 = decltype(p3_c)::value;
        const char* st = smem + (kt & 1) * BSTAGE;
        const char* stn = smem + ((kt + 1) & 1) * BSTAGE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int h = 0; h < 4; ++h) { af[(ks + 1) & 1][h] = frag(st, arow, h, ks + 1); bf[(ks + 1) & 1][h] = frag(st, brow, h, ks + 1); }
            } else if constexpr (BAR) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // tile kt+1 landed; every read of tile kt landed too
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int h = 0; h < 4; ++h) { af[0][h] = frag(stn, arow, h, 0); bf[0][h] = frag(stn, brow, h, 0); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    const int m = i * 4 + j;
                    if constexpr (P0) { if (ks == 0 && (m & 1)) issue_piece(kt + 1, 8 + (m >> 1)); }
                    if constexpr (P3) { if (ks == 3 && (m & 1)) issue_piece(kt + 2, m >> 1); }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::integral_constant<bool, true>; using F_ = std::integral_constant<bool, false>;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt + 2 < nkt; ++kt) tile(T_{}, T_{}, T_{}, kt);
    tile(T_{}, T_{}, F_{}, nkt - 2);
    tile(F_{}, F_{}, F_{}, nkt - 1);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wid == 0) cyc[blockIdx.x] = t1 - t0;
    float* Cb = C + (long)(blockIdx.x & 1) * BIG * BIG;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, col = wn * 128 + j * 32 + l31;
                Cb[row * BIG + col] = acc[i][j][r];
            }
}

template <int SKIP>
double run_big(const half_t* dA, float* dC, unsigned long long* dcyc, int K, int grid, int reps, double* cyc_per_kt) {
    auto kern = SKIP == 4 ? gemm_tile_big3<false> : SKIP == 3 ? gemm_tile_big3<true> : SKIP == 2 ? gemm_tile_big2<0> : gemm_tile_big<(SKIP == 1)>;   // 2 = slim issue, 3 = + early barrier, 4 = early barrier, nothing loaded
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BSTAGE);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 2 * BSTAGE, 0, dA, dC, dcyc, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 2 * BSTAGE, 0, dA, dC, dcyc, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    *cyc_per_kt = s / grid / (K / BK);
    return ms / reps * 1e3;
}

template <int NLOAD, int PRIO, int SKIP = 0, int SKEW = 0>
double run(const half_t* dA, float* dC, unsigned long long* dcyc, int K, int grid, int reps, double* cyc_per_kt) {
    auto kern = gemm_tile<NLOAD, PRIO, SKIP, SKEW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3((8 + NLOAD) * 64), NST * STAGE, 0, dA, dA, dC, dcyc, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3((8 + NLOAD) * 64), NST * STAGE, 0, dA, dA, dC, dcyc, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    *cyc_per_kt = s / grid / (K / BK);
    return ms / reps * 1e3;   // us
}

int main() {
    const int KMAX = 4096;
    std::vector<half_t> hA((size_t)512 * KMAX);            // 512 rows: enough for the 256 x 256 kernel's A-then-B image too
    srand(1);
    for (auto& v : hA) v = (half_t)(((rand() % 2001) - 1000) / 4000.0f);
    half_t* dA; float* dC; unsigned long long* dcyc;
    (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dC, 2 * 256 * 256 * 4); (void)hipMalloc(&dcyc, 4096 * 8);
    std::vector<float> hC(BM * BN);
    // ---- correctness at K = 256 (rows of the image are K apart: re-pack for this K) ----
    {
        const int K = 256;
        std::vector<half_t> a((size_t)(BM + BN) * K);
        for (int r = 0; r < BM + BN; ++r) for (int k = 0; k < K; ++k) a[(size_t)r * K + k] = hA[(size_t)r * KMAX + k];
        (void)hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice);
        std::vector<double> ref(BM * BN);
        for (int m = 0; m < BM; ++m) for (int n = 0; n < BN; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)a[(size_t)m * K + k] * (double)(float)a[(size_t)(BM + n) * K + k]; ref[m * BN + n] = s; }
        double cpk;
        auto check = [&](const char* nm) {
            (void)hipMemcpy(hC.data(), dC, BM * BN * 4, hipMemcpyDeviceToHost);
            double e = 0, r = 0; for (int i = 0; i < BM * BN; ++i) { e += (hC[i] - ref[i]) * (hC[i] - ref[i]); r += ref[i] * ref[i]; }
            printf("check %-22s K=256: rel-L2 vs fp64 host reference %.2e\n", nm, std::sqrt(e / r));
        };
        (void)hipMemset(dC, 0, 2 * BM * BN * 4); run<0, 0>(dA, dC, dcyc, K, 2, 1, &cpk); check("all waves load");
        (void)hipMemset(dC, 0, 2 * BM * BN * 4); run<0, 0, 0, 1>(dA, dC, dcyc, K, 2, 1, &cpk); check("skewed barrier");
        (void)hipMemset(dC, 0, 2 * BM * BN * 4); run<4, 0>(dA, dC, dcyc, K, 2, 1, &cpk); check("4 loader waves");
        (void)hipMemset(dC, 0, 2 * BM * BN * 4); run<4, 1>(dA, dC, dcyc, K, 2, 1, &cpk); check("4 loader waves, prio");
    }
    {
        const int K = 256;
        std::vector<half_t> a((size_t)512 * K);
        for (int r = 0; r < 512; ++r) for (int k = 0; k < K; ++k) a[(size_t)r * K + k] = hA[(size_t)r * KMAX + k];
        (void)hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice);
        double cpk; (void)hipMemset(dC, 0, 2 * 256 * 256 * 4); run_big<0>(dA, dC, dcyc, K, 2, 1, &cpk);
        std::vector<float> c(256 * 256); (void)hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
        double e = 0, rr = 0;
        for (int m = 0; m < 256; ++m) for (int n = 0; n < 256; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)a[(size_t)m * K + k] * (double)(float)a[(size_t)(256 + n) * K + k]; e += (c[m * 256 + n] - s) * (c[m * 256 + n] - s); rr += s * s; }
        printf("check %-22s K=256: rel-L2 vs fp64 host reference %.2e\n", "256x256, 4 waves", std::sqrt(e / rr));
        (void)hipMemset(dC, 0, 2 * 256 * 256 * 4); run_big<2>(dA, dC, dcyc, K, 2, 1, &cpk);
        (void)hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
        e = 0; rr = 0;
        for (int m = 0; m < 256; ++m) for (int n = 0; n < 256; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)a[(size_t)m * K + k] * (double)(float)a[(size_t)(256 + n) * K + k]; e += (c[m * 256 + n] - s) * (c[m * 256 + n] - s); rr += s * s; }
        printf("check %-22s K=256: rel-L2 vs fp64 host reference %.2e\n", "256x256, slim issue", std::sqrt(e / rr));
        (void)hipMemset(dC, 0, 2 * 256 * 256 * 4); run_big<3>(dA, dC, dcyc, K, 2, 1, &cpk);
        (void)hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
        e = 0; rr = 0;
        for (int m = 0; m < 256; ++m) for (int n = 0; n < 256; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)(float)a[(size_t)m * K + k] * (double)(float)a[(size_t)(256 + n) * K + k]; e += (c[m * 256 + n] - s) * (c[m * 256 + n] - s); rr += s * s; }
        printf("check %-22s K=256: rel-L2 vs fp64 host reference %.2e\n", "256x256, early barrier", std::sqrt(e / rr));
    }
    (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    printf("%-30s %6s %10s %10s %16s %12s\n", "variant", "WGs", "us", "TFLOP/s", "cycles / K-tile", "MFMA util");
    for (int grid : {1, 256, 1024}) {
        const int K = KMAX, reps = grid == 1 ? 5 : 20;
        const double flops = 2.0 * BM * BN * K * grid;
        double cpk, us;
        us = run<0, 0>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "all waves load", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<0, 0, 0, 1>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "skewed barrier", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<0, 0, 2, 1>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "skewed, nothing loaded", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<4, 0>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "4 loader waves", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<4, 1>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "4 loader waves, prio 3", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<0, 0, 1>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "all waves load, A only", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        { const double fb = 2.0 * 256 * 256 * K * grid;      // 256 x 256 tile: 64 MFMAs = 2048 matrix cycles per K-tile and SIMD (one wave)
          us = run_big<0>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "256x256, 4 waves of 128x128", grid, us, fb / us / 1e6, cpk, 2048.0 / cpk);
          us = run_big<2>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "256x256, slim piece issue", grid, us, fb / us / 1e6, cpk, 2048.0 / cpk);
          us = run_big<3>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "256x256, slim + early barrier", grid, us, fb / us / 1e6, cpk, 2048.0 / cpk);
          us = run_big<4>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "256x256, early bar., no loads", grid, us, fb / us / 1e6, cpk, 2048.0 / cpk);
          us = run_big<1>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "256x256, nothing loaded", grid, us, fb / us / 1e6, cpk, 2048.0 / cpk); }
        us = run<0, 0, 2>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "nothing loaded", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
        us = run<4, 0, 2>(dA, dC, dcyc, K, grid, reps, &cpk); printf("%-30s %6d %10.1f %10.1f %16.1f %12.3f\n", "4 idle loader waves", grid, us, flops / us / 1e6, cpk, 1024.0 / cpk);
    }
    printf("# MFMA util = 1024 matrix cycles per K-tile and SIMD (2 waves x 16 x 32) / measured shader cycles per K-tile (wave 0 of each workgroup)\n");
    return 0;
}
