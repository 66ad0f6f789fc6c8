// Device-side helpers shared by the implicit-GEMM kernels (igemm_kernel.hip, big4_kernel.hip): timeline stamps, exact
// small-quotient division, the epilogue-parameter segments in LDS, and every epilogue (plain, LDS-staged plain store,
// GEGLU, head-major Q / K / V^T).  Header-only, anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "igemm.h"

namespace {


// diagnostics: wave 0 / lane 0 of a workgroup stamps slot `slot` of its timeline record (IGemmArgs::tl, normally null)
__device__ __forceinline__ void tl_stamp(unsigned long long* tl, int slot) {
    if (tl != nullptr && threadIdx.x == 0) tl[(long)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void tl_begin(unsigned long long* tl) {
    if (tl != nullptr && threadIdx.x == 0) {
        unsigned long long* r = tl + (long)blockIdx.x * 16;
        r[0] = __builtin_amdgcn_s_memtime();
        r[4] = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r[6] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
}
__device__ __forceinline__ void tl_end(unsigned long long* tl) {
    if (tl != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores have left
        if (threadIdx.x == 0) {
            unsigned long long* r = tl + (long)blockIdx.x * 16;
            r[3] = __builtin_amdgcn_s_memtime();
            r[5] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// floor(m / d), m >= 0, for SMALL quotients (batch indices, image rows: < 2^20): one multiply by the hardware reciprocal and a
// one-step correction instead of the ~35-instruction integer-division sequence (no integer divider on the VALU).  The
// approximation is off by less than 0.4 for quotients below 2^20 even with a 2-ulp reciprocal, so one correction is exact
// (swept on the CPU over boundary cases for every divisor class the engines use).
__device__ __forceinline__ int qdiv(int m, int d) {
    // (the divisor is made opaque HERE: left alone, the compiler hoists the reciprocals of H*W and W to the top of the kernel and
    //  keeps them in VGPRs across the K loop, which pushed the 256-VGPR 256 x 320 kernel into 100 spills)
    int dv = d;
    asm volatile("" : "+v"(dv));
    const float inv = __builtin_amdgcn_rcpf((float)dv);
    int q = (int)((float)m * inv);
    const int r = m - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return q;
}

__device__ __forceinline__ int padded_pix(int m, int HW, int W, int H) {
    const int b = qdiv(m, HW), p = m - b * HW;
    const int y = qdiv(p, W), x = p - y * W;
    return (b * (H + 2) + y + 1) * (W + 2) + x + 1;
}


// workgroup index (XCD-contiguous numbering) -> tile.  Everything here is wave-uniform: the quotients go back to SGPRs
// (readfirstlane), so the blocked form costs no vector registers beyond the prologue.
__device__ __forceinline__ int qdiv_u(int a, int d) { return __builtin_amdgcn_readfirstlane(qdiv(a, d)); }
__device__ __forceinline__ void tile_of(const IGemmArgs& p, int wg, int& tile_m, int& tile_n) {
    if (p.walk_bn > 0) {                                   // XCD-blocked 2-D walk (IGemmArgs::walk_bn)
        const int x = qdiv_u(wg, p.walk_per), i = wg - x * p.walk_per;
        const int bmi = qdiv_u(x, p.walk_bn), bni = x - bmi * p.walk_bn;
        int im, in;
        if (p.n_major) { in = qdiv_u(i, p.walk_tmb); im = i - in * p.walk_tmb; }
        else { im = qdiv_u(i, p.walk_tnb); in = i - im * p.walk_tnb; }
        tile_m = bmi * p.walk_tmb + im; tile_n = bni * p.walk_tnb + in;
    } else {
        // (one division by a launcher-provided divisor: walk_div = ntn (M-major) or ntm (N-major))
        const int wq = qdiv(wg, p.walk_div), wr = wg - wq * p.walk_div;
        tile_m = p.n_major ? wr : wq; tile_n = p.n_major ? wq : wr;
    }
}

// ---- epilogue parameters in LDS -----------------------------------------------------------------------------------------
// The per-column parameters of an epilogue (bias, per-batch time embedding, LayerNorm column sums) used to be read from
// global memory where they are consumed: one 16-byte load per 4 columns, each inside its own `if (p.bias)` block and therefore
// followed by its own s_waitcnt vmcnt(0) - 25 (128x160 tile) to 60 (256x320 tile) serialised memory round trips per
// workgroup, 10-40 thousand cycles: the round-3 timelines show the epilogues costing 5-22 us per workgroup for that reason,
// not for bandwidth.  Now the tile's BN-column segments of those vectors go HBM -> LDS by LDS-DMA (4 bytes per lane) as the
// FIRST loads of the kernel: they are older than every K-tile piece, so the counted vmcnt of the K loop covers them, the
// K loop's barriers publish them, and the epilogue reads them with ds_read_b128.
constexpr int PAR_NB = 5;                                  // time-embedding rows (batches) of a tile when H*W >= 64 (BM <= 256): the sizes the
                                                           // occupancy figures assume; smaller feature maps take IGemmArgs::par_nb rows
__host__ __device__ constexpr int par_bnp(int BN) { return (BN + 63) / 64 * 64; }
__host__ __device__ constexpr int par_bytes(int BN, int nb = PAR_NB) { return par_bnp(BN) * 4 * (1 + nb); }
struct Par {
    const char* lds;     // null: read the parameters from global memory (register-staged kernels, the K-split reduce kernel)
    int n0, b0, bnp;     // first column / first batch of the tile, padded segment length (floats)
};
// issue the DMA pieces (64 floats each) of the tile's parameter segments; NW = waves of the workgroup
template <int BN, int NW>
__device__ __forceinline__ void par_stage(const IGemmArgs& p, char* par, int n0, int m0, int wid, int lane) {
    constexpr int BNP = par_bnp(BN), NPC = BNP / 64;
    int nn = n0 + lane;                                     // + 64 * piece, clamped per piece
    const int HW = p.rows_per_batch;
    const int b0 = HW > 0 ? qdiv(m0, HW) : 0;
    const int nb = HW > 0 ? (p.M + HW - 1) / HW : 1;
    int pi = 0;                                             // running piece index -> wave pi % NW (compile-time after unrolling)
    auto arr = [&](const float* src, int slot) {
#pragma unroll
        for (int q = 0; q < NPC; ++q, ++pi) {
            if (wid == pi % NW) {
                int n = nn + 64 * q;
                n = n < p.N ? n : p.N - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + n),
                                                 (__attribute__((address_space(3))) void*)(par + (slot * BNP + q * 64) * 4), 4, 0, 0);
            }
        }
    };
    if (p.bias) arr(p.bias, 0);
    if (p.temb) {
        for (int k = 0; k < p.par_nb; ++k) {               // (run-time count: 2 .. 5 for H*W >= 64)
            int b = b0 + k;
            b = b < nb ? b : nb - 1;
            arr(p.temb + (long)b * p.temb_ld, 1 + k);
        }
    }
}
// L = the kernel staged the segments in LDS (compile-time: a run-time choice kept a global-load path with its waits alive)
template <bool L>
__device__ __forceinline__ float4 par_bias4(const IGemmArgs& p, const Par& q, int n) {
    if constexpr (L) return *reinterpret_cast<const float4*>(q.lds + (n - q.n0) * 4);
    else return *reinterpret_cast<const float4*>(p.bias + n);
}
template <bool L>
__device__ __forceinline__ float4 par_temb4(const IGemmArgs& p, const Par& q, int b, int n) {
    if constexpr (L) {
        int k = b - q.b0;                              // < par_nb by the launcher's choice of par_nb
        k = k < p.par_nb ? k : p.par_nb - 1;
        k = k < 0 ? 0 : k;                             // (par_nb == 0: no batch structure - row 0)
        return *reinterpret_cast<const float4*>(q.lds + ((1 + k) * q.bnp + n - q.n0) * 4);
    } else {
        return *reinterpret_cast<const float4*>(p.temb + (long)b * p.temb_ld + n);
    }
}
// PIN (big4_kernel: 256 accumulator registers in the AGPR half of the file): keeps row block i of the accumulators in AGPRs until
// the epilogue gets to it.  Without it the compiler copies ALL accumulators to VGPRs (v_accvgpr_read x 192) above the branch
// between the epilogue variants and spills the rest to scratch.
template <bool PIN, int NT>
__device__ __forceinline__ void acc_row_pin(f32x16 (&row)[NT]) {
    if constexpr (PIN) {
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+a"(row[j]));
    }
}
// Shared epilogue.  lane: m = mw0 + i*32 + (lane&31);  n = nw0 + j*32 + 8*g + 4*(lane>>5) + {0..3}, g = reg>>2
template <int MT, int NT, bool L>
__device__ __forceinline__ void igemm_epilogue(const IGemmArgs& p, f32x16 (&acc)[MT][NT], int mw0, int nw0, int lane, const Par& par) {
    const int frow = lane & 31, fhi = lane >> 5;
    const int HW = p.rows_per_batch;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mw0 + i * 32 + frow;
        if (m >= p.M) continue;
        const int b = (HW > 0) ? qdiv(m, HW) : 0;
        const int tok = m - b * HW;
        long orow = m, rrow = m;
        if (p.omode == 1 || p.rmode == 1) {
            const long pp = padded_pix(m, HW, p.W, p.H);
            if (p.omode == 1) orow = pp;
            if (p.rmode == 1) rrow = pp;
        }
        if (p.epi == EPI_STORE) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int n = nw0 + j * 32 + 8 * g + 4 * fhi;
                    // opaque per (i, j, g): keeps the compiler from hoisting ~50 column pointers out of the i loop
                    // (they were spilled to scratch on the 10-accumulator-tile config)
                    asm volatile("" : "+v"(n));
                    if (n >= p.N) continue;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k] * p.out_scale;
                    if (p.bias) {
                        const float4 bb = par_bias4<L>(p, par, n);
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    if (p.temb) {
                        const float4 tt = par_temb4<L>(p, par, b, n);
                        v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
                    }
                    if (p.resid) {
                        const half4_t rr = *reinterpret_cast<const half4_t*>(p.resid + rrow * p.rld + n);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += (float)rr[k];
                    }
                    half4_t o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                    *reinterpret_cast<half4_t*>(p.out + orow * p.old + n) = o;
                }
        } else if (p.epi == EPI_GEGLU) {
            // packed columns: within every 64 packed columns, [0,32) = value, [32,64) = gate
            if constexpr (NT % 2 == 0) {
#pragma unroll
                for (int j = 0; j < NT; j += 2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int pc = nw0 + j * 32 + 8 * g + 4 * fhi;   // packed value column
                        if (pc >= p.N) continue;
                        const int f = (pc >> 6) * 32 + (pc & 31);
                        float v[4], gt[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { v[k] = acc[i][j][4 * g + k]; gt[k] = acc[i][j + 1][4 * g + k]; }
                        if (p.bias) {
                            const float4 bv = par_bias4<L>(p, par, pc);
                            const float4 bg = par_bias4<L>(p, par, pc + 32);
                            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                            gt[0] += bg.x; gt[1] += bg.y; gt[2] += bg.z; gt[3] += bg.w;
                        }
                        half4_t o;
                        const f32x2 g01 = gelu_erf_pk((f32x2){gt[0], gt[1]}), g23 = gelu_erf_pk((f32x2){gt[2], gt[3]});
                        o[0] = (half_t)(v[0] * g01.x); o[1] = (half_t)(v[1] * g01.y);
                        o[2] = (half_t)(v[2] * g23.x); o[3] = (half_t)(v[3] * g23.y);
                        *reinterpret_cast<half4_t*>(p.out + orow * p.old + f) = o;
                    }
            }
        } else {  // EPI_HEADS
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int n = nw0 + j * 32 + 8 * g + 4 * fhi;
                    asm volatile("" : "+v"(n));
                    if (n >= p.N) continue;
                    const int part = n / p.part_width + p.part0;
                    const int cn = n % p.part_width;
                    const int head = cn / p.head_dim, dd = cn - head * p.head_dim;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k];
                    if (p.bias) {
                        const float4 bb = par_bias4<L>(p, par, n);
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    const long bh = (long)b * p.heads + head;
                    if (part == 2) {
                        half_t* dst = p.hvt + (bh * p.head_dim_pad + dd) * p.tok_pad + (p.vt_linear ? tok : cfgpp_vt_pos(tok));
#pragma unroll
                        for (int k = 0; k < 4; ++k) dst[(long)k * p.tok_pad] = (half_t)v[k];
                    } else {
                        half4_t o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                        half_t* base = part == 0 ? p.hq : p.hk;
                        const int tp = part == 0 ? p.q_tok_pad : p.tok_pad;
                        *reinterpret_cast<half4_t*>(base + (bh * tp + tok) * p.head_dim_pad + dd) = o;
                    }
                }
        }
    }
}

// GroupNorm statistics of one staged 32-row block (IGemmArgs::gstat): lane cp takes the column pair (2 cp, 2 cp + 1) of the
// wave's WTN-column slab in LDS (fp16, the values that were just stored), pivot = the pair's values in row 0, rows 1 .. 31 in
// order: (mean, M2) per column, written as one float4 per pair.  The arithmetic per (block, column) is the same whatever tile
// config staged the block, so the statistics - like the outputs - do not depend on tile tuning.
template <int WTN, int PITCH>
__device__ __forceinline__ void gstat_block(const IGemmArgs& p, const char* stg, int m_blk0, int nw0, int lane) {
    if (m_blk0 >= p.M) return;
    float* dst = p.gstat + ((long)(m_blk0 >> 5) * p.N + nw0) * 2;
    // (rolled loops on purpose: unrolled, the 31 LDS reads of a pass are hoisted into 31 more live registers, which the tiles that
    //  sit at their register budget - 64 x 160 waves at 243 of 256, the four-waves-per-SIMD 256 x 128 tile at 126 of 128 - spill)
#pragma unroll 1
    for (int cp = lane; cp < WTN / 2; cp += 64) {
        const half2_t x0 = *reinterpret_cast<const half2_t*>(stg + cp * 4);
        const float pv0 = (float)x0[0], pv1 = (float)x0[1];
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll 4
        for (int r = 1; r < 32; ++r) {
            const half2_t x = *reinterpret_cast<const half2_t*>(stg + r * PITCH + cp * 4);
            const float d0 = (float)x[0] - pv0, d1 = (float)x[1] - pv1;
            s0 += d0; q0 += d0 * d0; s1 += d1; q1 += d1 * d1;
        }
        if (nw0 + 2 * cp < p.N)
            *reinterpret_cast<float4*>(dst + 4 * cp) = make_float4(pv0 + s0 * (1.0f / 32.0f), q0 - s0 * s0 * (1.0f / 32.0f),
                                                                   pv1 + s1 * (1.0f / 32.0f), q1 - s1 * s1 * (1.0f / 32.0f));
    }
}

// EPI_STORE through LDS: the MFMA accumulator layout gives each lane 4 consecutive features of ONE row,
// i.e. 8-byte stores scattered over 32 rows per instruction (and the same pattern for the residual
// read).  Here each wave transposes its 32 x WTN slab through a private LDS region and then moves whole
// row segments (WTN*2 bytes contiguous, 16 B per lane): coalesced residual loads and output stores.
// bias / time-embedding are added in fp32 before the (single) rounding to fp16; the residual is added to
// the fp16 value, which is exactly the reference's `conv(...)` (fp16) `+ residual` (fp16) order.
template <int MT, int NT, bool L, bool PIN = false>
__device__ __forceinline__ void igemm_epilogue_staged(const IGemmArgs& p, f32x16 (&acc)[MT][NT], int mw0, int nw0, int lane,
                                                      char* stg /* wave-private, 32 * (NT*64 + 16) bytes */, const Par& par) {
    constexpr int WTN = NT * 32;
    constexpr int PITCH = WTN * 2 + 16;
    constexpr int CPR = WTN / 8;                 // 16-B chunks per row
    constexpr int NQ = (32 * CPR + 63) / 64;
    // the residual pieces of a 32-row slab are requested in ONE block of loads (clamped addresses, no per-piece branch: a
    // branch per piece made every load wait for itself), BEHIND the transpose through LDS: every read of the parameter
    // segments carries a compiler-inserted vmcnt(0) (they were written by LDS-DMA), which is free only while no other load is
    // in flight.  One memory latency per 32-row slab stays exposed.
    constexpr bool EARLY = false;
    const int frow = lane & 31, fhi = lane >> 5;
    const int HW = p.rows_per_batch;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc_row_pin<PIN, NT>(acc[i]);
        const int m = mw0 + i * 32 + frow;
        const int mc = m < p.M ? m : p.M - 1;
        const int b = (HW > 0) ? qdiv(mc, HW) : 0;
        // this lane's row: output / residual pixel index (shared with the other lanes by shuffle below)
        int opix = mc, rpix = mc;
        if (p.omode == 1 || p.rmode == 1) {
            const int pp = padded_pix(mc, HW, p.W, p.H);
            if (p.omode == 1) opix = pp;
            if (p.rmode == 1) rpix = pp;
        }
        half8_t rres[NQ];
        auto request_residual = [&]() {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c = lane + 64 * q;
                const int r = c / CPR, cc = c - r * CPR;
                const int rp = __shfl(rpix, r);                 // (rows >= 32 of a padded last pass wrap to a valid row)
                int n = nw0 + cc * 8;
                n = n < p.N ? n : p.N - 8;
                rres[q] = *reinterpret_cast<const half8_t*>(p.resid + (long)rp * p.rld + n);
            }
        };
        if constexpr (EARLY) { if (p.resid) request_residual(); }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * fhi;
                int n = nw0 + nl;
                n = n < p.N ? n : p.N - 4;          // clamp (values unused beyond N)
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k] * p.out_scale;
                if (p.bias) {
                    const float4 bb = par_bias4<L>(p, par, n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.temb) {
                    const float4 tt = par_temb4<L>(p, par, b, n);
                    v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
                }
                half4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                *reinterpret_cast<half4_t*>(stg + frow * PITCH + nl * 2) = o;
            }
        if constexpr (!EARLY) { if (p.resid) request_residual(); }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = lane + 64 * q;
            const int r = c / CPR, cc = c - r * CPR;
            const int op = __shfl(opix, r);                      // row r's output pixel index (lane r holds row r)
            const int mm = mw0 + i * 32 + r, n = nw0 + cc * 8;
            half8_t v = *reinterpret_cast<const half8_t*>(stg + (r & 31) * PITCH + cc * 16);
            if (p.resid) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)rres[q][k]);
                if (p.gstat && c < 32 * CPR) *reinterpret_cast<half8_t*>(stg + (r & 31) * PITCH + cc * 16) = v;      // the stored value, for the statistics below
            }
            if (c < 32 * CPR && mm < p.M && n < p.N) *reinterpret_cast<half8_t*>(p.out + (long)op * p.old + n) = v;
        }
        if (p.gstat) gstat_block<WTN, PITCH>(p, stg, mw0 + i * 32, nw0, lane);
    }
}

// GEGLU through LDS: value tile j and gate tile j+1 of a wave hold the same 32 features; the product
// v * gelu(g) is staged as [32 rows][NT/2*32 features] and written as row segments (16 B per lane).
template <int MT, int NT, bool L, bool PIN = false>
__device__ __forceinline__ void igemm_epilogue_geglu_staged(const IGemmArgs& p, f32x16 (&acc)[MT][NT], int mw0, int nw0, int lane,
                                                            char* stg, const Par& par) {
    constexpr int WTF = (NT / 2) * 32;            // output features per wave
    constexpr int PITCH = WTF * 2 + 16;
    constexpr int CPR = WTF / 8;
    constexpr int NQ = (32 * CPR + 63) / 64;
    const int frow = lane & 31, fhi = lane >> 5;
    const int f0 = (nw0 >> 6) * 32;               // packed column -> feature (nw0 is a multiple of 64)
    const int NF = p.N >> 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc_row_pin<PIN, NT>(acc[i]);
#pragma unroll
        for (int j = 0; j < NT; j += 2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int pc = nw0 + j * 32 + 8 * g + 4 * fhi;          // packed value column
                pc = pc < p.N ? pc : p.N - 64;
                float v[4], gt[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[i][j][4 * g + k]; gt[k] = acc[i][j + 1][4 * g + k]; }
                if (p.bias) {
                    const float4 bv = par_bias4<L>(p, par, pc);
                    const float4 bg = par_bias4<L>(p, par, pc + 32);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                    gt[0] += bg.x; gt[1] += bg.y; gt[2] += bg.z; gt[3] += bg.w;
                }
                half4_t o;
                const f32x2 g01 = gelu_erf_pk((f32x2){gt[0], gt[1]}), g23 = gelu_erf_pk((f32x2){gt[2], gt[3]});
                o[0] = (half_t)(v[0] * g01.x); o[1] = (half_t)(v[1] * g01.y);
                o[2] = (half_t)(v[2] * g23.x); o[3] = (half_t)(v[3] * g23.y);
                *reinterpret_cast<half4_t*>(stg + frow * PITCH + ((j >> 1) * 32 + 8 * g + 4 * fhi) * 2) = o;
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = lane + 64 * q;
            const int r = c / CPR, cc = c - r * CPR;
            const int mm = mw0 + i * 32 + r, f = f0 + cc * 8;
            if (c < 32 * CPR && mm < p.M && f < NF)
                *reinterpret_cast<half8_t*>(p.out + (long)mm * p.old + f) = *reinterpret_cast<const half8_t*>(stg + r * PITCH + cc * 16);
        }
    }
}

// Head-major addressing without per-piece integer divisions.  A runtime-divisor division is a ~35-instruction VALU sequence;
// the staged heads epilogues did two per 16-byte piece plus one per group in each loop (52 per wave on the 256 x 256 tile:
// ~15 000 of the ~34 000 epilogue cycles the round-3 timeline shows for the QKV projections).  The first column of an
// aligned column group is wave-uniform, so (part, head, offset in the head) are computed ONCE per group and a piece
// `off` columns further (off < 32 <= head_dim) is at most one conditional wrap away.
struct HeadCol { int part, head, dd; };
__device__ __forceinline__ HeadCol head_col(const IGemmArgs& p, int ng_uniform) {
    const int ng = __builtin_amdgcn_readfirstlane(ng_uniform);
    const int pr = ng / p.part_width, cn0 = ng - pr * p.part_width;
    const int h = cn0 / p.head_dim;
    HeadCol c; c.part = pr + p.part0; c.head = h; c.dd = cn0 - h * p.head_dim;
    return c;
}
__device__ __forceinline__ void head_step(const IGemmArgs& p, const HeadCol& c, int off, int& head, int& dd) {
    dd = c.dd + off; head = c.head;
    if (p.head_dim >= 32) {
        if (dd >= p.head_dim) { dd -= p.head_dim; ++head; }
    } else {                                                   // (head dims under 32: test-sized models only)
        const int q = dd / p.head_dim;
        head += q; dd -= q * p.head_dim;
    }
}

// EPI_HEADS through LDS.  The plain epilogue above scatters 8-byte pieces (Q, K) and - for V^T, which is stored
// transposed - single halves (4 two-byte stores per lane and accumulator group): measured, the QKV projection ran at
// 664 TF/s where the same GEMM with a plain store runs at 930.  Here every 32-token x 32-column accumulator sub-tile is
// staged in a 32 x (64 + 16)-byte LDS block - row-major [token][column] for Q / K columns, TRANSPOSED
// [column = head dim][token position] for V columns (positions = the attention kernel's permuted key order) - and
// leaves as 16-byte pieces: 8 consecutive head dims of one token (Q, K) or 8 consecutive key positions of one head
// dim (V^T).  Needs rows_per_batch % 32 == 0 (a sub-tile then lies inside one batch and one 32-key block) and
// part_width % 32 == 0, head_dim % 8 == 0 (a 32-column group has one part, a 16-byte piece one head).
template <int MT, int NT, bool L, bool PIN = false>
__device__ __forceinline__ void igemm_epilogue_heads_staged(const IGemmArgs& p, f32x16 (&acc)[MT][NT], int mw0, int nw0, int lane,
                                                            char* stg /* wave-private, NT * 2560 bytes */, const Par& par) {
    constexpr int PITCH = 80, BLK = 32 * PITCH;
    const int frow = lane & 31, fhi = lane >> 5;
    const int HW = p.rows_per_batch;
    const int ppos = cfgpp_vt_pos(frow);                       // this lane's token -> key position inside its 32-block
    HeadCol hc[NT];                                            // per 32-column group: part / head / offset of its first column
#pragma unroll
    for (int j = 0; j < NT; ++j) { const int ng = nw0 + j * 32; hc[j] = head_col(p, ng < p.N ? ng : p.N - 32); }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        acc_row_pin<PIN, NT>(acc[i]);
        const int m0s = __builtin_amdgcn_readfirstlane(mw0 + i * 32);   // first token row of the sub-tile (multiple of 32)
        if (m0s >= p.M) continue;
        const int b = qdiv(m0s, HW), tok0 = m0s - b * HW;           // one batch, one aligned 32-token block
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ng = nw0 + j * 32;                       // first column of the 32-column group
            const int part = hc[j].part;
            char* blk = stg + j * BLK;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                int n = ng + 8 * g + 4 * fhi;
                n = n < p.N ? n : p.N - 4;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k];
                if (p.bias) {
                    const float4 bb = par_bias4<L>(p, par, n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (part == 2 && !p.vt_linear) {               // transposed: [head dim column][key position]
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<half_t*>(blk + (8 * g + 4 * fhi + k) * PITCH + ppos * 2) = (half_t)v[k];
                } else if (part == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<half_t*>(blk + (8 * g + 4 * fhi + k) * PITCH + frow * 2) = (half_t)v[k];
                } else {
                    half4_t o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                    *reinterpret_cast<half4_t*>(blk + frow * PITCH + (8 * g + 4 * fhi) * 2) = o;
                }
            }
        }
        // 32 rows x 4 pieces of 16 B per 32-column group: 2 pieces per lane
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ng = nw0 + j * 32;
            if (ng >= p.N) continue;
            const int part = hc[j].part;
            const char* blk = stg + j * BLK;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = lane + 64 * q;
                const int r = c >> 2, c4 = c & 3;
                const half8_t v = *reinterpret_cast<const half8_t*>(blk + r * PITCH + c4 * 16);
                if (part == 2) {                               // row r = head-dim column ng + r, piece = 8 key positions
                    int head, dd;
                    head_step(p, hc[j], r, head, dd);
                    const long bh = (long)b * p.heads + head;
                    if (ng + r < p.N)
                        *reinterpret_cast<half8_t*>(p.hvt + (bh * p.head_dim_pad + dd) * p.tok_pad + tok0 + 8 * c4) = v;
                } else {                                       // row r = token tok0 + r, piece = 8 head dims
                    const int n = ng + 8 * c4;
                    int head, dd;
                    head_step(p, hc[j], 8 * c4, head, dd);
                    const long bh = (long)b * p.heads + head;
                    half_t* base = part == 0 ? p.hq : p.hk;
                    const int tp = part == 0 ? p.q_tok_pad : p.tok_pad;
                    if (n < p.N)
                        *reinterpret_cast<half8_t*>(base + (bh * tp + tok0 + r) * p.head_dim_pad + dd) = v;
                }
            }
        }
    }
}

}  // namespace
