// One-wave-per-SIMD implicit-GEMM tiles (big4_kernel.hip): configs 24 .. 26.  The launcher (igemm_kernel.hip) prepares the
// argument block (tile walk, parameter rows, grid) and calls big4_run.
#pragma once
#include "igemm.h"

inline void big4_tile(int cfg, int* BM, int* BN) {
    *BM = cfg == 24 ? 256 : 128;
    *BN = cfg == 25 ? 320 : 256;
}
// Admission test: the kernels have LDS-staged epilogues only, 32-bit byte offsets against a scalar base (both operands must
// span < 4 GiB; pixel indices and row pitches < 2^24 for the 24-bit multiply), and the 320-wide tile has no GEGLU pairs.
inline bool big4_supports(int cfg, const IGemmArgs& a, bool staged_epi) {
    if (cfg < 24 || cfg > 26 || !staged_epi || a.K < 64) return false;
    const long rows = a.amode == 0 ? (long)a.M : a.amode == 2 ? (long)(a.M / a.rows_per_batch) * (2 * a.H + 2) * (2 * a.W + 2)
                                                : (long)(a.M / a.rows_per_batch) * (a.H + 2) * (a.W + 2);      // (amode 3 reads a smaller map)
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    if (rows >= (1L << 24) || rows * cmax * 2 >= (1L << 32) || (long)a.N * a.K * 2 >= (1L << 32) || a.N >= (1 << 24)) return false;
    if (a.epi == EPI_STORE) return (a.N & 7) == 0;
    if (a.epi == EPI_GEGLU) return cfg != 25 && (a.N & 127) == 0 && a.omode == 0;
    return a.epi == EPI_HEADS && (a.rows_per_batch & 31) == 0 && (a.part_width & 31) == 0 && (a.head_dim & 7) == 0 && (a.N & 31) == 0 &&
           (a.M & 31) == 0;
}
int big4_par_bytes(int BN, int nb);      // LDS bytes of the epilogue-parameter segments behind the ring (nb time-embedding rows)
int big4_run(int cfg, const IGemmArgs& a, int grid, int smem, hipStream_t stream);

// config 28 (big4p_kernel.hip): the 256 x 256 tile as a PERSISTENT kernel for token-major linears (amode 0, one source): one
// workgroup per CU walks its share of the output tiles and prefetches the next tile's first K-tile under the epilogue.
inline bool big4p_supports(const IGemmArgs& a, bool staged_epi) {
    return a.amode == 0 && a.C1 == 0 && a.taps == 1 && a.temb == nullptr && big4_supports(24, a, staged_epi) && (long)a.K * 2 < (1L << 24);
}
int big4p_smem(int par_bytes_one);       // ring + two parameter-row buffers
int big4p_run(const IGemmArgs& a, int grid, int smem, int par_bytes_one, hipStream_t stream);
