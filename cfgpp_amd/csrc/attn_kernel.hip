// K6/K7: fused (flash-style) attention for the SD / SDXL UNet on gfx950.
//
//   O[b, q, h*d + :] = softmax(Q K^T * scale) V      Q,K: [B*heads][tok_pad][dp]   V^T: [B*heads][dp][tok_pad]
//
// Layout contract (written by the QKV GEMM epilogue, EPI_HEADS): head-major, head
// dim zero-padded to dp = round_up(d, 32), V stored TRANSPOSED with the keys of every
// 32-key block PERMUTED (bits 2 and 3 of the key index swapped: cfgpp_vt_pos in common.h) so
// that every MFMA operand is ONE contiguous 16-byte LDS read and no cross-lane shuffles are needed
// (the 8 k-slots a lane feeds to a PV MFMA are its accumulator registers 8*tt .. 8*tt+7 = keys
// 16*tt + 8*b + 4*hi + r, which the permutation makes consecutive):
//   * S^T = K Q^T with v_mfma_f32_32x32x16_f16 (A = K rows = keys, B = Q^T cols =
//     queries): each lane ends up holding 16 scores of ONE query per 32-key tile.
//   * Online softmax entirely in registers (one __shfl_xor(…,32) for the row max); Q is pre-scaled by
//     d^-1/2*log2(e) so scores feed v_exp_f32 directly; the O rescale is skipped (exactly) on tiles
//     where no query's running max grows; key masking only on a partial last tile.
//   * The loop is VALU-bound, not MFMA-bound, at small d (150 VALU vs 14 MFMA per 64-key tile at d = 40), so
//     two thirds of the softmax VALU is moved into the MFMAs: (i) the accumulator of S^T is INITIALISED with
//     -m (a persistent 16-register vector, rewritten only when a running max grows), so scores come out as
//     s - m and feed v_exp_f32 without a subtract; (ii) when d is not a multiple of 32 the first padding row
//     of V^T holds ONES (written once at engine build, cfgpp_op_attention_prepare_vt), so row d of O^T
//     accumulates sum_k P[k][q] - the softmax denominator - inside the PV MFMAs (no v_add chain, and the
//     denominator sums exactly the fp16 P values the numerator uses).
//   * O^T += V^T P^T: the B operand (P^T) is exactly the lane's own 8 consecutive
//     accumulator registers converted to fp16 (the MFMA k-slot <-> key assignment is
//     free as long as A and B agree), the A operand is one ds_read_b128 from the permuted V^T.
//   * 4 waves x 32 queries per workgroup, 64-key tiles.
//   * dp = 64 (d = 40: SD1.5 64x64 level, d = 64: every SDXL level - > 90 % of the attention time):
//     attn64_kernel.  K and V^T tiles are both [64 rows][128 B]; they go HBM -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: no VGPR staging, no ds_write pass - the VGPR -> LDS store path was what
//     saturated the CU's LDS pipe: PMC showed 33 % of the LDS cycles as bank conflicts of the old 2 x
//     ds_read_b64 V reads and 47 % of the wave time parked) into unpadded rows with the igemm's XOR swizzle
//     applied to the per-lane SOURCE chunk, on a 2-stage ring (32 KB: FOUR workgroups per CU since round 4; rounds 2-3: 3 stages, three), one
//     raw s_barrier per tile, counted vmcnt.
//   * other head dims (80, 160): attn_kernel, register-staged double buffering, padded LDS rows.
// Cross-attention (77 keys padded to 128) uses the same kernel with nk_valid = 77.
// softmax statistics and accumulation are fp32; P is rounded to fp16 for the PV
// product (same as the reference's SDPA flash path under fp16 autocast).
#include "common.h"

namespace {

// Deferred re-referencing of the online softmax: the scores leave the MFMA as s - m_ref (log2 domain); as long as no
// query of the wave exceeds m_ref by more than RESCALE_THR the reference stays and P = 2^(s - m_ref) <= 2^THR = 32
// (exact in fp16 / fp32: a floating-point scale; O and the denominator carry the same factor and it cancels in O / l).
// On random data a NEW running max appears in ~2/3 of the tiles of a 32-query wave, but it beats the old one by < 2;
// re-referencing only on a jump > 5 removes ~50 VALU instructions from 2/3 of the tiles.  The rare branch is covered
// by tests/test_gpu_configs.py::test_attention_online_softmax_rescale_branch (a key that jumps by ~100).
constexpr float RESCALE_THR = 5.0f;

struct AttnArgs {
    const half_t* q; const half_t* k; const half_t* vt;
    half_t* o;
    int heads, d;         // real head dim
    int nq, nk_valid;     // real query / key counts
    int q_tok_pad, k_tok_pad;
    int o_ld;             // heads*d
    int nqb;              // query blocks (of 128 queries) per batch*head; grid = nqb * B * heads workgroups
    float scale_log2e;    // d^-0.5 * log2(e)
    int stagger;          // > 0: the co-resident workgroups of a CU start phase-shifted by this many 64-cycle sleeps (A/B knob)
};

// Workgroup -> (query block, batch*head).  Workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never
// correctness), and every XCD has its own 4 MiB L2.  All query blocks of one (batch, head) read the same K / V^T
// (1 MB at N = 4096): with the natural order they are spread over all 8 XCDs and every L2 has to hold the K / V^T
// of EVERY head in flight (24 heads x 1 MB >> 4 MiB), so K / V^T tiles keep coming from HBM / Infinity Cache
// (~4 GB of fabric traffic per launch instead of 128 MB).  The remap hands each XCD a contiguous range of
// (batch*head, query block) pairs: the ~3 heads an XCD works on at a time stay L2-resident.
__device__ __forceinline__ void attn_block_map(int nqb, int& qb, int& bh) {
    const int T = gridDim.x, bid = blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    bh = w / nqb; qb = w - bh * nqb;
}

template <int D16, int DT, bool ONES, int QT>
__global__ void __launch_bounds__(256)
attn_kernel(const AttnArgs a) {
    constexpr int DP = DT * 32;                  // padded head dim (row pitch of Q/K, rows of V^T)
    constexpr int KPITCH = DP * 2 + 16;          // bytes per K row in LDS
    constexpr int VPITCH = 64 * 2 + 16;          // bytes per V^T row in LDS (64 keys)
    constexpr int CH = (64 * DP / 8) / 256;      // 16-B chunks per thread per tile (K and V each)
    constexpr int STAGE = 64 * KPITCH + DP * VPITCH;
    static_assert((64 * DP / 8) % 256 == 0, "tile/loader mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 2 stages of (K tile | V^T tile)

    // QT query sub-tiles of 32 per wave: every K / V^T fragment read from LDS feeds QT MFMAs.  Measured with
    // QT = 2 (240-256 VGPRs, 2 waves/SIMD): +2 % at d = 40, -11 % at d = 64 (N = 4096), so QT = 1 ships.
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int qb, bh;
    attn_block_map(a.nqb, qb, bh);
    const int q0 = qb * (128 * QT) + wid * (32 * QT);
    const half_t* Qb = a.q + (long)bh * a.q_tok_pad * DP;
    const half_t* Kb = a.k + (long)bh * a.k_tok_pad * DP;
    const half_t* Vb = a.vt + (long)bh * DP * a.k_tok_pad;

    // Q^T fragments (B operand): lane = query q0+qt*32+l31, 8 consecutive d at ks*16 + hi*8, pre-multiplied by
    // d^-1/2 * log2(e) so that the scores come out of the MFMA ready for exp2.
    half8_t qf[QT][D16];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ks = 0; ks < D16; ++ks) {
            const half8_t raw = *reinterpret_cast<const half8_t*>(Qb + (long)(q0 + qt * 32 + l31) * DP + ks * 16 + hi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[qt][ks][j] = (half_t)((float)raw[j] * a.scale_log2e);
        }

    f32x16 oacc[QT][DT];
    // negm = -m_ref broadcast over the 16 accumulator registers (C operand of the first S^T MFMA);
    // m_ref = 0 until the first tile (which always takes the "max grew" path) sets it.
    f32x16 negm[QT];
    float l_run[QT];                 // only used when !ONES
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        l_run[qt] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[qt][r] = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qt][i][r] = 0.f;
    }

    const int ntiles = (a.nk_valid + 63) >> 6;
    const int tail = a.nk_valid & 63;            // != 0: the last tile is partially masked

    half8_t rk[CH], rv[CH];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = tid + j * 256;
            const int kr = c / (DP / 8), kc = c - kr * (DP / 8);
            rk[j] = *reinterpret_cast<const half8_t*>(Kb + (long)(t * 64 + kr) * DP + kc * 8);
            const int vr = c >> 3, vc = c & 7;
            rv[j] = *reinterpret_cast<const half8_t*>(Vb + (long)vr * a.k_tok_pad + t * 64 + vc * 8);
        }
    };
    auto store_tile = [&](int stage) {
        char* Ks = smem + stage * STAGE;
        char* Vs = Ks + 64 * KPITCH;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = tid + j * 256;
            const int kr = c / (DP / 8), kc = c - kr * (DP / 8);
            *reinterpret_cast<half8_t*>(Ks + kr * KPITCH + kc * 16) = rk[j];
            const int vr = c >> 3, vc = c & 7;
            *reinterpret_cast<half8_t*>(Vs + vr * VPITCH + vc * 16) = rv[j];
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    if (ntiles > 1) load_tile(1);
    for (int t = 0; t < ntiles; ++t) {
        const char* Ks = smem + (t & 1) * STAGE;
        const char* Vs = Ks + 64 * KPITCH;

        // ---- S^T - m = K Q^T + (-m) for two 32-key sub-tiles (log2 domain) ----
        f32x16 s[QT][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int ks = 0; ks < D16; ++ks) {
                const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + (kt * 32 + l31) * KPITCH + (ks * 16 + hi * 8) * 2);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qt][ks], ks == 0 ? negm[qt] : s[qt][kt], 0, 0, 0);
            }
        }
        if (tail != 0 && t == ntiles - 1) {      // wave-uniform: mask keys >= nk_valid (cross-attention, 77 keys)
            const int kbase = t * 64 + 4 * hi;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
                        s[qt][kt][r] = key < a.nk_valid ? s[qt][kt][r] : -INFINITY;
                    }
        }
        // ---- online softmax: per query = per lane column; keys across 32 registers and the two half-waves ----
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = s[qt][0][0];                  // tile max RELATIVE to m_ref
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qt][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (t == 0 || !__all(mx <= RESCALE_THR)) {   // (rare after the first tile) re-reference: exact
                const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);         // O = 0 on the first tile: alpha irrelevant
                l_run[qt] *= alpha;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qt][i][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qt][r] -= delta;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qt][kt][r] -= delta;
            }
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[qt][kt][r]);
                    s[qt][kt][r] = pv;
                    if constexpr (!ONES) psum += pv;
                }
            if constexpr (!ONES) l_run[qt] += psum;
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                half8_t pf[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[qt][j] = (half_t)s[qt][kt][8 * tt + j];
                // k-slot j = b*4 + r  <->  key = kt*32 + 8*(2*tt + b) + 4*hi + r  <->  V^T column kt*32 + 16*tt + 8*hi + j
                const int kpos0 = kt * 32 + 16 * tt + 8 * hi;
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const half8_t vf = *reinterpret_cast<const half8_t*>(Vs + (i * 32 + l31) * VPITCH + kpos0 * 2);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        oacc[qt][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qt], oacc[qt][i], 0, 0, 0);
                }
            }
        // ---- stage tile t+1 (registers -> the other LDS stage), prefetch tile t+2 ----
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        __syncthreads();
        if (t + 2 < ntiles) load_tile(t + 2);
    }

    // ---- finalize: O = O^T / l, store token-major ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l_tot;
        if constexpr (ONES) {
            // row d of O^T (the ones row of V^T) = sum_k P[k][q]: register (d%32 / 8) * 4 of tile d/32, half-wave 0
            const int dr = a.d & 31;                                   // multiple of 8 by contract
            float lv = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) if (dr == 8 * g) lv = oacc[qt][DT - 1][4 * g];
            l_tot = __shfl(lv, l31);
        } else {
            l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
        }
        const float inv_l = 1.0f / l_tot;
        const int q = q0 + qt * 32 + l31;
        if (q < a.nq) {
            const int b = bh / a.heads, head = bh - b * a.heads;
            half_t* orow = a.o + ((long)b * a.nq + q) * a.o_ld + head * a.d;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dd = i * 32 + 8 * g + 4 * hi;
                    if (dd < a.d) {
                        half4_t o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (half_t)(oacc[qt][i][4 * g + k] * inv_l);
                        *reinterpret_cast<half4_t*>(orow + dd) = o;
                    }
                }
        }
    }
}

// ---- dp = 64 ------------------------------------------------------------------------------------
// K tile  : LDS [64 keys  ][64 halfs], row = key,   16-B chunk c = 8 head dims
// V^T tile: LDS [64 d-rows][64 keys ], row = d,     16-B chunk c = 8 (permuted) keys
// physical chunk = logical chunk ^ ((row >> 1) & 7); one DMA piece = 8 rows x 128 B = 64 lanes x 16 B.
template <int D16, bool ONES, int NST, int WPE = 3>        // NST = LDS ring stages (shipped: 2 = 32 KB); WPE = waves per SIMD the
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))      // register allocation targets (4: <= 128 VGPRs, with NST = 2 four workgroups / CU)
attn64_kernel(const AttnArgs a) {
    constexpr int DP = 64, DT = 2;
    constexpr int TILE = 64 * 128;               // bytes of one K or V^T tile
    constexpr int STAGE = 2 * TILE;
    constexpr int PPW = 4;                       // DMA pieces per wave per tile: 8 (K) + 8 (V^T) over 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int qb, bh;
    attn_block_map(a.nqb, qb, bh);
    const int q0 = qb * 128 + wid * 32;
    const half_t* Qb = a.q + (long)bh * a.q_tok_pad * DP;
    const half_t* Kb = a.k + (long)bh * a.k_tok_pad * DP;
    const half_t* Vb = a.vt + (long)bh * DP * a.k_tok_pad;

    // DMA source addressing: lane -> (row r8 = lane >> 3 of the piece, physical chunk pc = lane & 7); the wave's
    // pieces are K rows [16 * wid, 16 * wid + 16) and V^T rows [16 * wid, 16 * wid + 16) of the tile
    const int r8 = lane >> 3, pc = lane & 7;
    // (32-bit element offsets from the wave-uniform bases Kb / Vb: half the registers of per-lane 64-bit pointers, and the loads can
    // take the scalar-base + vector-offset form)
    int koff[2], voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = wid * 16 + h * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        koff[h] = row * DP + lc * 8;                               // + t * 64 * DP per tile
        voff[h] = row * a.k_tok_pad + lc * 8;                      // + t * 64 per tile
    }
    auto dma_tile = [&](int t, int stage) {
        char* Ks = smem + stage * STAGE;
        char* Vs = Ks + TILE;
        const half_t* Kt = Kb + (long)t * 64 * DP;                 // (wave-uniform)
        const half_t* Vt = Vb + (long)t * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kt + koff[h]),
                                             (__attribute__((address_space(3))) void*)(Ks + (wid * 16 + h * 8) * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vt + voff[h]),
                                             (__attribute__((address_space(3))) void*)(Vs + (wid * 16 + h * 8) * 128), 16, 0, 0);
        }
    };

    const int ntiles = (a.nk_valid + 63) >> 6;
    const int tail = a.nk_valid & 63;
    if (a.stagger > 0) {
        // Workgroups that share a CU run the same code from the same instant: their QK^T / softmax / PV phases coincide
        // and the matrix and vector pipes take turns instead of overlapping (PMC: both busy 19 % of the time).  Dispatch
        // slot s of a CU (observed order: block b -> XCD b % 8, then one block per CU, then the second slot ...) starts s
        // thirds of a tile late; blocks that start later inherit the offset of the block whose slot they take.
        const int slot = ((blockIdx.x >> 3) >> 5) % 3;
        for (int i = 0; i < slot * a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) if (t < ntiles) dma_tile(t, t);

    // Q^T fragments (B operand), pre-multiplied by d^-1/2 * log2(e); ordinary loads, issued after the first DMAs
    half8_t qf[D16];
#pragma unroll
    for (int ks = 0; ks < D16; ++ks) {
        const half8_t raw = *reinterpret_cast<const half8_t*>(Qb + (long)(q0 + l31) * DP + ks * 16 + hi * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)raw[j] * a.scale_log2e);
    }

    f32x16 oacc[DT];
    f32x16 negm;
    float l_run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;

    const int fsw = (l31 >> 1) & 7;
    const int frow = l31 * 128;
    for (int t = 0; t < ntiles; ++t) {
        // tile t landed once at most the NST-2 younger tiles' pieces are outstanding; the barrier makes every wave's pieces
        // visible and tells that everyone is done reading tile t-1, whose stage tile t+NST-1 is about to overwrite
        if (NST > 2 && t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");           // (the barrier intrinsic alone does not order the LDS reads below)
        if (t + NST - 1 < ntiles) dma_tile(t + NST - 1, (t + NST - 1) % NST);
        const char* Ks = smem + (t % NST) * STAGE;
        const char* Vs = Ks + TILE;

        // ---- S^T - m = K Q^T + (-m) for two 32-key sub-tiles (log2 domain) ----
        f32x16 s[2];
        {
            // register-lean form (<= 128 VGPRs: four waves per SIMD): one 32-key sub-tile's K fragments at a time; the other waves of
            // the SIMD cover the LDS latency
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int k0 = 0; k0 < D16; k0 += 2) {    // two fragments (8 registers) in flight at a time
                    half8_t kf1[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (k0 + u < D16) kf1[u] = *reinterpret_cast<const half8_t*>(Ks + kt * 32 * 128 + frow + (((((k0 + u) << 1) | hi) ^ fsw) << 4));
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (k0 + u < D16) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1[u], qf[k0 + u], k0 + u == 0 ? negm : s[kt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);   // later fragments are not hoisted above these MFMAs (registers)
                }
            }
        }
        if (tail != 0 && t == ntiles - 1) {      // wave-uniform: mask keys >= nk_valid (cross-attention, 77 keys)
            const int kbase = t * 64 + 4 * hi;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
                    s[kt][r] = key < a.nk_valid ? s[kt][r] : -INFINITY;
                }
        }
        // ---- online softmax: per query = per lane column; keys across 32 registers and the two half-waves ----
        float mx = s[0][0];                      // tile max RELATIVE to m_ref
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (t == 0 || !__all(mx <= RESCALE_THR)) {   // (rare after the first tile) re-reference: exact
            const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kt][r] -= delta;
        }
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kt][r]);
                s[kt][r] = pv;
                if constexpr (!ONES) psum += pv;
            }
        if constexpr (!ONES) l_run += psum;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                half8_t pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)s[kt][8 * tt + j];
                const int lc = kt * 4 + 2 * tt + hi;               // logical 16-B chunk of the (permuted) key axis
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const half8_t vf = *reinterpret_cast<const half8_t*>(Vs + i * 32 * 128 + frow + ((lc ^ fsw) << 4));
                    oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[i], 0, 0, 0);
                }
            }
    }

    // ---- finalize: O = O^T / l, store token-major ----
    // (lane-derived indices are re-derived here from an opaque copy of the lane id: kept live across the key loop they are what
    //  the register-lean instantiations spilled)
    int lane_f = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_f));
    const int l31_f = lane_f & 31, hi_f = lane_f >> 5;
    float l_tot;
    if constexpr (ONES) {
        const int dr = a.d & 31;
        float lv = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) if (dr == 8 * g) lv = oacc[DT - 1][4 * g];
        l_tot = __shfl(lv, l31_f);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv_l = 1.0f / l_tot;
    const int q = qb * 128 + wid * 32 + l31_f;
    if (q < a.nq) {
        const int b = bh / a.heads, head = bh - b * a.heads;
        half_t* orow = a.o + ((long)b * a.nq + q) * a.o_ld + head * a.d;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dd = i * 32 + 8 * g + 4 * hi_f;
                if (dd < a.d) {
                    half4_t o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (half_t)(oacc[i][4 * g + k] * inv_l);
                    *reinterpret_cast<half4_t*>(orow + dd) = o;
                }
            }
    }
}

// ---- cross-attention, dp = 64, <= 128 keys (the 77 text tokens) ----------------------------------------------------------
// With 77 keys the flash loop above is two tiles of latency per 128 queries: every workgroup loads its own copy of K / V^T
// (32 KB) for 16 KB of Q and 16 KB of O, waits for it twice and runs the online-softmax bookkeeping for nothing
// (100-160 TF/s in the per-launch tables of rounds 1-2).  Here a workgroup loads the head's K and V^T ONCE (both 64-key tiles, by
// LDS-DMA, the layouts of attn64_kernel) and then walks XQB blocks of 128 queries with no barrier and no running maximum: all
// 96 scores of a query (three 32-key sub-tiles; keys >= nk_valid masked) are in registers, softmax is a single pass, and the
// kernel is what it should be - a stream of Q in and O out.
template <int D16, bool ONES>
__global__ void __launch_bounds__(256)
xattn64_kernel(const AttnArgs a, int xqb /* 128-query blocks per workgroup */, int nxb /* workgroups per batch*head */) {
    constexpr int DP = 64, DT = 2, TILE = 64 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // K tile 0 | V^T tile 0 | K tile 1 | V^T tile 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int xb, bh;
    {   // XCD-contiguous (batch*head, chunk) order as attn_block_map
        const int T = gridDim.x, bid = blockIdx.x;
        const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bh = w / nxb; xb = w - bh * nxb;
    }
    const half_t* Qb = a.q + (long)bh * a.q_tok_pad * DP;
    const half_t* Kb = a.k + (long)bh * a.k_tok_pad * DP;
    const half_t* Vb = a.vt + (long)bh * DP * a.k_tok_pad;
    const int ntiles = (a.nk_valid + 63) >> 6;               // 1 or 2
    {
        const int r8 = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < ntiles) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = wid * 16 + h * 8 + r8;
                    const int lc = pc ^ ((row >> 1) & 7);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kb + (long)(t * 64 + row) * DP + lc * 8),
                                                     (__attribute__((address_space(3))) void*)(smem + t * 2 * TILE + (wid * 16 + h * 8) * 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vb + (long)row * a.k_tok_pad + t * 64 + lc * 8),
                                                     (__attribute__((address_space(3))) void*)(smem + t * 2 * TILE + TILE + (wid * 16 + h * 8) * 128), 16, 0, 0);
                }
            }
        }
    }
    const int fsw = (l31 >> 1) & 7;
    const int frow = l31 * 128;
    const int nsub = (a.nk_valid + 31) >> 5;                 // 32-key sub-tiles with at least one valid key (1 .. 4)
    const int b = bh / a.heads, head = bh - b * a.heads;
    // first block's Q while the DMA is in flight
    half8_t qf[D16];
    int q0 = (xb * xqb) * 128 + wid * 32;
#pragma unroll
    for (int ks = 0; ks < D16; ++ks) qf[ks] = *reinterpret_cast<const half8_t*>(Qb + (long)(q0 + l31) * DP + ks * 16 + hi * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int blk = 0; blk < xqb; ++blk) {
        half8_t qs[D16];
#pragma unroll
        for (int ks = 0; ks < D16; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) qs[ks][j] = (half_t)((float)qf[ks][j] * a.scale_log2e);
        const int qrow = q0 + l31;
        // next block's Q (clamped: the last block re-reads itself)
        const int q0n = blk + 1 < xqb ? q0 + 128 : q0;
#pragma unroll
        for (int ks = 0; ks < D16; ++ks) qf[ks] = *reinterpret_cast<const half8_t*>(Qb + (long)(q0n + l31) * DP + ks * 16 + hi * 8);
        // ---- S^T = K Q^T for up to four 32-key sub-tiles ----
        f32x16 s[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
            if (st < nsub) {
                const char* Ks = smem + (st >> 1) * 2 * TILE + (st & 1) * 32 * 128;
#pragma unroll
                for (int ks = 0; ks < D16; ++ks) {
                    const half8_t kf = *reinterpret_cast<const half8_t*>(Ks + frow + ((((ks << 1) | hi) ^ fsw) << 4));
                    s[st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qs[ks], s[st], 0, 0, 0);
                }
            }
        }
        // ---- softmax over all keys at once (log2 domain); keys >= nk_valid masked ----
        float mx = -INFINITY;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st < nsub) {
                if (st == nsub - 1 && (a.nk_valid & 31) != 0) {      // (wave-uniform) only the last sub-tile has invalid keys
                    const int nv = (a.nk_valid & 31) - 4 * hi;         // valid keys of this sub-tile, seen from this half-wave
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[st][r] = ((r & 3) + 8 * (r >> 2)) < nv ? s[st][r] : -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[st][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float psum = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st < nsub) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[st][r] - mx);
                    s[st][r] = pv;
                    if constexpr (!ONES) psum += pv;
                }
            }
        }
        // ---- O^T = V^T P^T ----
        f32x16 oacc[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st < nsub) {
                const char* Vs = smem + (st >> 1) * 2 * TILE + TILE;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    half8_t pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (half_t)s[st][8 * tt + j];
                    const int lc = (st & 1) * 4 + 2 * tt + hi;
#pragma unroll
                    for (int i = 0; i < DT; ++i) {
                        const half8_t vf = *reinterpret_cast<const half8_t*>(Vs + i * 32 * 128 + frow + ((lc ^ fsw) << 4));
                        oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[i], 0, 0, 0);
                    }
                }
            }
        }
        float l_tot;
        if constexpr (ONES) {
            const int dr = a.d & 31;
            float lv = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) if (dr == 8 * g) lv = oacc[DT - 1][4 * g];
            l_tot = __shfl(lv, l31);
        } else {
            l_tot = psum + __shfl_xor(psum, 32);
        }
        const float inv_l = 1.0f / l_tot;
        if (qrow < a.nq) {
            half_t* orow = a.o + ((long)b * a.nq + qrow) * a.o_ld + head * a.d;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dd = i * 32 + 8 * g + 4 * hi;
                    if (dd < a.d) {
                        half4_t o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (half_t)(oacc[i][4 * g + k] * inv_l);
                        *reinterpret_cast<half4_t*>(orow + dd) = o;
                    }
                }
        }
        q0 = q0n;
    }
}

template <int D16, bool ONES, int NST, int WPE = 3>
int launch_attn64(const AttnArgs& a, dim3 grid, hipStream_t s) {
    constexpr int smem = NST * 2 * 64 * 128;
    static bool attr_set = false;
    auto kern = attn64_kernel<D16, ONES, NST, WPE>;
    if (!attr_set) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    return 0;
}

// sets row d of every V^T matrix to 1.0 (see the kernel header); vt [BH][dp][tok_pad]
__global__ void attn_ones_row_kernel(half_t* vt, int BH, int d, int dp, int tok_pad) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)BH * tok_pad) return;
    const long bh = i / tok_pad, k = i - bh * tok_pad;
    vt[(bh * dp + d) * tok_pad + k] = (half_t)1.0f;
}

template <int D16, int DT, bool ONES, int QT>
int launch_attn(const AttnArgs& a, dim3 grid, hipStream_t s) {
    constexpr int DP = DT * 32;
    constexpr int smem = 2 * (64 * (DP * 2 + 16) + DP * (64 * 2 + 16));
    static bool attr_set = false;
    auto kern = attn_kernel<D16, DT, ONES, QT>;
    if (!attr_set) {
        CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
    return 0;
}

}  // namespace

static int g_attn_dma = 1;       // dp = 64: 1 = LDS-DMA kernel, 0 = register-staged kernel (A/B switch).  (A software-pipelined form
                                 // with two score tiles live was built and measured in round 3 - correct, 8 % slower: fewer resident waves -
                                 // and removed in round 4; profiles/r03/ab/attention_variants_alone.txt.  Round 4's woven form - QK^T of
                                 // tile t+1 and PV of tile t-1 issued between slices of tile t's softmax, two workgroups per CU - was +6..11 %
                                 // alone, neutral per forward and no better in matrix-pipe utilisation: commit 926a7ad)
static int g_attn_cross = 1;     // dp = 64, <= 128 keys: 1 = the resident-K/V cross-attention kernel, 0 = the flash loop (A/B switch)
static int g_attn_stagger = 0;   // attn64_kernel: phase shift between the workgroups of a CU, in 64-cycle sleeps per slot (0 = off)

extern "C" {

void cfgpp_attention_set_dma(int mode) { g_attn_dma = mode ? 1 : 0; }
void cfgpp_attention_set_stagger(int sleeps) { g_attn_stagger = sleeps > 0 ? sleeps : 0; }
void cfgpp_attention_set_cross(int on) { g_attn_cross = on ? 1 : 0; }

// V^T contract: when d is not a multiple of 32, row d of every [dp][tok_pad] matrix must hold ones (softmax
// denominator through the PV MFMA).  Call once after allocating / zeroing the buffer; the QKV epilogue never
// writes rows >= d.
int cfgpp_op_attention_prepare_vt(void* vt, int BH, int d, int tok_pad, void* stream) {
    CFGPP_REQUIRE(vt && BH > 0 && d > 0 && tok_pad > 0, "attention_prepare_vt: bad args");
    if (d % 32 == 0) return 0;
    CFGPP_REQUIRE(d % 8 == 0, "attention: head dim %d (not a multiple of 32) must be a multiple of 8", d);
    const int dp = (d + 31) / 32 * 32;
    hipLaunchKernelGGL(attn_ones_row_kernel, dim3(cdiv((long)BH * tok_pad, 256)), dim3(256), 0, (hipStream_t)stream, (half_t*)vt, BH, d, dp, tok_pad);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

// q [B*heads][q_tok_pad][dp], k [B*heads][k_tok_pad][dp], vt [B*heads][dp][k_tok_pad], all fp16, dp = round_up(d,32),
// pads zero EXCEPT the ones row of vt (above).  o [B][nq][heads*d] fp16.  q_tok_pad % 128 == 0, k_tok_pad % 64 == 0.
int cfgpp_op_attention(const void* q, const void* k, const void* vt, void* o, int B, int heads, int d,
                       int nq, int nk, int q_tok_pad, int k_tok_pad, void* stream) {
    CFGPP_REQUIRE(q && k && vt && o, "attention: null pointer");
    CFGPP_REQUIRE(d > 0 && d <= 160 && (d % 32 == 0 || d % 8 == 0), "attention: head dim %d unsupported (multiple of 8, <= 160)", d);
    CFGPP_REQUIRE(q_tok_pad % 128 == 0 && q_tok_pad >= nq, "attention: q_tok_pad=%d (nq=%d) must be a multiple of 128", q_tok_pad, nq);
    CFGPP_REQUIRE(k_tok_pad % 64 == 0 && k_tok_pad >= nk, "attention: k_tok_pad=%d (nk=%d) must be a multiple of 64", k_tok_pad, nk);
    AttnArgs a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.vt = (const half_t*)vt; a.o = (half_t*)o;
    a.heads = heads; a.d = d; a.nq = nq; a.nk_valid = nk; a.q_tok_pad = q_tok_pad; a.k_tok_pad = k_tok_pad;
    a.o_ld = heads * d;
    a.scale_log2e = (1.0f / sqrtf((float)d)) * 1.4426950408889634f;
    a.nqb = cdiv(nq, 128);
    a.stagger = g_attn_stagger;
    dim3 grid(a.nqb * B * heads);
    hipStream_t s = (hipStream_t)stream;
    const int d16 = (d + 15) / 16, dt = (d + 31) / 32;
    const bool ones = (d % 32) != 0;
    if (dt == 2 && g_attn_dma && g_attn_cross && nk <= 128 && k_tok_pad >= 64 * ((nk + 63) >> 6)) {
        // cross-attention (77 text tokens): K / V^T resident per workgroup, single-pass softmax, XQB query blocks per workgroup
        const int BH = B * heads;
        int xqb = 1;
        while (xqb < 8 && a.nqb % (xqb * 2) == 0 && (long)BH * (a.nqb / (xqb * 2)) >= 512) xqb *= 2;
        const int nxb = a.nqb / xqb;
        static bool attr_set = false;
        if (!attr_set) {
            CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128));
            CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128));
            CFGPP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128));
            attr_set = true;
        }
        const dim3 xg(BH * nxb);
        if (d16 == 3) hipLaunchKernelGGL((xattn64_kernel<3, true>), xg, dim3(256), 4 * 64 * 128, s, a, xqb, nxb);
        else if (ones) hipLaunchKernelGGL((xattn64_kernel<4, true>), xg, dim3(256), 4 * 64 * 128, s, a, xqb, nxb);
        else hipLaunchKernelGGL((xattn64_kernel<4, false>), xg, dim3(256), 4 * 64 * 128, s, a, xqb, nxb);
        CFGPP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (dt == 2 && g_attn_dma) {                   // dp = 64 (d = 40, 48, 56, 64): LDS-DMA kernel
        int rc;        // (a 2-stage ring was measured within 1 % of the 3-stage one and is not built)
        // four workgroups per CU: two-stage ring (32 KB), <= 128 VGPRs (the three-per-CU form on a 3-stage ring of rounds 2-3 measured
        // 3 - 9 % slower alone and 0.3 - 1 % per forward, profiles/r04/ab/attention_occupancy_call9.txt / _call10.txt)
        if (d16 == 3) rc = launch_attn64<3, true, 2, 4>(a, grid, s);
        else rc = ones ? launch_attn64<4, true, 2, 4>(a, grid, s) : launch_attn64<4, false, 2, 4>(a, grid, s);
        if (rc) return -1;
        CFGPP_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define ATTN_CASE(D16_, DT_) \
    if (d16 == D16_ && dt == DT_) { if (ones ? launch_attn<D16_, DT_, true, 1>(a, grid, s) : launch_attn<D16_, DT_, false, 1>(a, grid, s)) return -1; } else
    ATTN_CASE(1, 1) ATTN_CASE(2, 1) ATTN_CASE(3, 2) ATTN_CASE(4, 2) ATTN_CASE(5, 3) ATTN_CASE(6, 3)
    ATTN_CASE(8, 4) ATTN_CASE(10, 5)
    { cfgpp_set_error("attention: no kernel instance for head dim %d", d); return -2; }
#undef ATTN_CASE
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
