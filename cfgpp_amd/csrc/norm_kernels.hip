// K8 GroupNorm(32)(+SiLU) and K9 LayerNorm for the SD/SDXL UNet (HBM-bound).
//
// GroupNorm works on halo-padded NHWC fp16 and can read its input as the channel
// concatenation of two tensors (the skip-connection `torch.cat([h, skip], 1)` of
// diffusers' up blocks never materialises).  Statistics are fp32 (autocast keeps
// group_norm / layer_norm in fp32); the normalised, affine, optionally SiLU'd
// value is rounded to fp16 once - exactly the value the following fp16 conv /
// linear consumes in the reference.
//
// Two forms, both deterministic (no atomics, fixed summation order -> bit-identical runs) and both
// free of the E[x^2] - mean^2 cancellation (activations with |mean| >> sigma, e.g. what the fp16-fix
// VAE exists for):
//  (1) SLAB kernel, ONE launch, 4 B / element (one read, one write): a workgroup owns all H*W pixels
//      of one sample for a set of gs consecutive groups (gs*C/G channels = 80 ... 240 B per pixel) and
//      holds that slab in REGISTERS (the 512 KB register file of a CU is its largest memory); mean and
//      variance are the exact two-pass form (sum, then sum of squared deviations), like torch.
//      Used whenever the slab fits 16 x 16-B chunks per thread of a <= 960-thread workgroup (H*W <= 1024 at
//      every channel count of the two UNets): all 32x32 / 16x16 / 8x8 GroupNorms of SD1.5 and the 32x32 ones of SDXL.
//  (2) two launches for larger slabs (64x64 and 128x128 levels): per-block partial sums of (x - pivot),
//      (x - pivot)^2 with a per-(sample, group) pivot = the group's first element, then an apply kernel
//      whose prologue reduces the partials in a fixed order.  6 B / element.
#include <algorithm>
#include "common.h"

namespace {

struct GNArgs {
    const half_t* src0;
    const half_t* src1;   // may be null when C1 == 0
    half_t* dst;
    const float* gamma;   // [C]
    const float* beta;    // [C]
    float* stats;         // [N][nblk][G][2] fp32 per-block partial (sum, sumsq) of (x - pivot)
    int nblk;
    int N, H, W;
    int C0, C1;           // channels of src0 / src1 (C = C0 + C1), both multiples of 8
    int G;                // groups
    float eps;
    int silu;
    int pix_per_block;
    int dst_padded;       // 1: dst is halo-padded NHWC, 0: dst is token-major [N*H*W][C]
    int gs;               // slab kernel: groups per workgroup
    // statistics from the PRODUCERS of src0 / src1 (igemm epilogues, IGemmArgs::gstat): per 32-pixel block and channel {mean, M2};
    // gn_finalize_kernel folds them into stats[(n * G + g) * 2] = {mean, rstd}, which gn_apply_kernel then takes as given (pre = 1)
    const float* gst0; const float* gst1;
    int pre;
};

__device__ __forceinline__ long pad_off(int n, int y, int x, int H, int W) {
    return ((long)(n * (H + 2) + y + 1) * (W + 2) + x + 1);
}

// pivot of (sample n, group g): the group's first channel at the sample's first pixel (any value within a few
// sigma of the group mean removes the cancellation; this one costs one L2 hit)
__device__ __forceinline__ float gn_pivot(const GNArgs& a, int n, int c_first) {
    const long p = pad_off(n, 0, 0, a.H, a.W);
    return c_first < a.C0 ? (float)a.src0[p * a.C0 + c_first] : (float)a.src1[p * a.C1 + (c_first - a.C0)];
}

// Makes the packed fp16 slab registers opaque between the passes: otherwise the compiler converts every element
// to fp32 once and keeps all of them live (2x the registers; spills at 16+ chunks per thread).
__device__ __forceinline__ void gn_opaque(half8_t& v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u = __builtin_bit_cast(u32x4, v);
    asm volatile("" : "+v"(u));
    v = __builtin_bit_cast(half8_t, u);
}

// ---- (1) slab kernel ------------------------------------------------------------------------------
// grid.x = N * (G / gs) workgroups (XCD-contiguous order: neighbouring channel ranges of one sample share an
// L2, so the 128-B lines their 80..240-B segments straddle are fetched from HBM once).  NT = cpp * R threads,
// cpp = 16-B chunks per pixel segment: thread t owns chunk column cc = t % cpp (its 8 channels, hence its
// group(s), scale and shift are loop-invariant) of pixels t / cpp + j * R.
template <int MAXCH, int NT>
__global__ void __launch_bounds__(NT)
gn_slab_kernel(GNArgs a) {
    __shared__ float s_part[2][NT];          // per-thread partials (lower / upper group of the thread's chunk)
    __shared__ float s_col[2][32];           // per chunk-column totals
    __shared__ float s_grp[4];               // per local group: mean, then rstd
    const int C = a.C0 + a.C1, cpg = C / a.G;
    const int chw = a.gs * cpg, cpp = chw >> 3;
    const int S = a.G / a.gs;
    const int HW = a.H * a.W;
    // XCD-contiguous remap of the workgroup index (bijective form)
    const int T = gridDim.x, bid = blockIdx.x;
    const int q = T >> 3, r8 = T & 7, xcd = bid & 7, idx = bid >> 3;
    const int w = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int n = w / S, sset = w - n * S;
    const int tid = threadIdx.x;
    const int R = NT / cpp;                  // pixel rows per pass (NT is a multiple of cpp by construction)
    const int prow = tid / cpp, cc = tid - prow * cpp;
    const int c = sset * chw + cc * 8;       // first channel of this thread's chunk (global channel index)
    const half_t* src; int cs, Cs;
    if (c < a.C0) { src = a.src0; cs = c; Cs = a.C0; } else { src = a.src1; cs = c - a.C0; Cs = a.C1; }
    // local groups of the chunk: channels [cc*8, cc*8+8) of the workgroup's range; at most two (cpg >= 8)
    const int g_lo = (cc * 8) / cpg;
    const int split = min(8, (g_lo + 1) * cpg - cc * 8);       // elements [0, split) belong to g_lo, the rest to g_lo + 1
    const float invW = 1.0f / (float)a.W;
    const long pbase = (long)n * (a.H + 2) * (a.W + 2) + (a.W + 2) + 1;
    auto poff = [&](int p) -> long { return pbase + p + 2 * (int)(((float)p + 0.5f) * invW); };

    half8_t v[MAXCH];
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int p = prow + j * R;
        if (p < HW) v[j] = *reinterpret_cast<const half8_t*>(src + poff(p) * Cs + cs);
        else { for (int k = 0; k < 8; ++k) v[j][k] = (half_t)0.f; }
    }
    // fixed-order block reduction of (lo, hi) per-thread partials into per-group totals:
    // s_part -> one wave per chunk column (shuffle tree over the R rows) -> s_col -> thread g sums its columns
    const int lane = tid & 63, wid = tid >> 6, nw = NT >> 6;
    auto block_reduce = [&](float lo, float hi, float* out4 /* s_grp */) {
        s_part[0][cc * R + prow] = lo; s_part[1][cc * R + prow] = hi;
        __syncthreads();
        for (int col = wid; col < cpp; col += nw) {
            float x = 0.f, y = 0.f;
            for (int rr = lane; rr < R; rr += 64) { x += s_part[0][col * R + rr]; y += s_part[1][col * R + rr]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
            if (lane == 0) { s_col[0][col] = x; s_col[1][col] = y; }
        }
        __syncthreads();
        if (tid < a.gs) {
            float t = 0.f;
            for (int col = 0; col < cpp; ++col) {
                const int gl = (col * 8) / cpg;
                if (gl == tid) t += s_col[0][col];
                if (gl + 1 == tid) t += s_col[1][col];
            }
            out4[tid] = t;
        }
        __syncthreads();
    };
    const float inv_cnt = 1.0f / ((float)cpg * (float)HW);
    // pass 1: mean
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float f = (float)v[j][k]; if (k < split) lo += f; else hi += f; }
    block_reduce(lo, hi, s_grp);
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) gn_opaque(v[j]);
    const float mean_lo = s_grp[g_lo] * inv_cnt;
    const float mean_hi = (split < 8) ? s_grp[g_lo + 1] * inv_cnt : 0.f;
    __syncthreads();                         // s_grp is rewritten by pass 2
    // pass 2: sum of squared deviations (padding chunks beyond HW hold zeros: exclude them)
    lo = 0.f; hi = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        if (prow + j * R < HW) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = (float)v[j][k];
                if (k < split) { const float d = f - mean_lo; lo += d * d; } else { const float d = f - mean_hi; hi += d * d; }
            }
        }
    }
    block_reduce(lo, hi, s_grp);
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) gn_opaque(v[j]);
    const float rstd_lo = rsqrtf(s_grp[g_lo] * inv_cnt + a.eps);
    const float rstd_hi = (split < 8) ? rsqrtf(s_grp[g_lo + 1] * inv_cnt + a.eps) : 0.f;
    // apply
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float m = k < split ? mean_lo : mean_hi, rs = k < split ? rstd_lo : rstd_hi;
        const float ga = a.gamma[c + k] * rs;
        sc[k] = ga; sh[k] = a.beta[c + k] - m * ga;
    }
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        int p = prow + j * R;
        asm volatile("" : "+v"(p));          // recompute the pixel offset here instead of keeping MAXCH 64-bit addresses live
        if (p < HW) {
            half8_t o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float f = (float)v[j][k] * sc[k] + sh[k];
                if (a.silu) f = silu_f(f);
                o[k] = (half_t)f;
            }
            const long orow = a.dst_padded ? poff(p) : ((long)n * HW + p);
            *reinterpret_cast<half8_t*>(a.dst + orow * C + c) = o;
        }
    }
}

// ---- (2) two-launch form ---------------------------------------------------------------------------
// Pass 1: per-(sample, pixel-block) partial sums of (x - pivot), (x - pivot)^2 for every group: each thread sums
// 8 channels over its pixels, the block combines them through LDS in a fixed order and writes part[n][blk][g][2].
__global__ void __launch_bounds__(256)
gn_stats_kernel(GNArgs a) {
    __shared__ float s_part[256][16];   // [thread][8 sums | 8 sums of squares]
    const int C = a.C0 + a.C1;
    const int cpg = C / a.G;
    const int chunks = C / 8;
    const int n = blockIdx.y;
    const int HW = a.H * a.W;
    const int p0 = blockIdx.x * a.pix_per_block;
    const int p1 = min(p0 + a.pix_per_block, HW);
    const int ppi = max(1, (int)blockDim.x / chunks);        // pixels per iteration
    // padded offset of pixel p of sample n without an integer division: pad_off = base + p + 2*(p / W), and
    // p / W = floor((p + 0.5) / W) is exact in fp32 here (p < 2^20, H <= 1024: error 1e-4 << margin 0.5 / W)
    const float invW = 1.0f / (float)a.W;
    const long pbase = (long)n * (a.H + 2) * (a.W + 2) + (a.W + 2) + 1;
    auto poff = [&](int p) -> long { return pbase + p + 2 * (int)(((float)p + 0.5f) * invW); };
    float gsum = 0.f, gsq = 0.f;                             // thread g < G owns group g
    for (int cbase = 0; cbase < chunks; cbase += blockDim.x) {
        int chunk, psub;
        if (chunks <= (int)blockDim.x) { chunk = threadIdx.x % chunks; psub = threadIdx.x / chunks; }
        else { chunk = cbase + threadIdx.x; psub = 0; }
        const bool active = (chunk < chunks) && (psub < ppi);
        float s[8], q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
        if (active) {
            const int c = chunk * 8;
            const half_t* src; int cs, Cs;
            if (c < a.C0) { src = a.src0; cs = c; Cs = a.C0; } else { src = a.src1; cs = c - a.C0; Cs = a.C1; }
            float pv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) pv[k] = gn_pivot(a, n, ((c + k) / cpg) * cpg);
            // 4 independent 16-B loads in flight per thread; the per-thread summation order is unchanged
            int p = p0 + psub;
            for (; p + 3 * ppi < p1; p += 4 * ppi) {
                half8_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8_t*>(src + poff(p + u * ppi) * Cs + cs);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float f = (float)v[u][k] - pv[k]; s[k] += f; q[k] += f * f; }
            }
            for (; p < p1; p += ppi) {
                const half8_t v = *reinterpret_cast<const half8_t*>(src + poff(p) * Cs + cs);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float f = (float)v[k] - pv[k]; s[k] += f; q[k] += f * f; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s_part[threadIdx.x][k] = s[k]; s_part[threadIdx.x][8 + k] = q[k]; }
        __syncthreads();
        if ((int)threadIdx.x < a.G) {
            // channels of group g inside this pass: [max(g*cpg, cbase*8), min((g+1)*cpg, (cbase+blockDim)*8, C))
            const int g = threadIdx.x;
            const int pass_lo = cbase * 8, pass_hi = min(C, (cbase + (int)blockDim.x) * 8);
            const int c_lo = max(g * cpg, pass_lo), c_hi = min((g + 1) * cpg, pass_hi);
            for (int c = c_lo; c < c_hi; ++c) {
                const int chunk_l = (c >> 3) - cbase, k = c & 7;
                if (chunks <= (int)blockDim.x) {
                    for (int ps = 0; ps < ppi; ++ps) { gsum += s_part[ps * chunks + chunk_l][k]; gsq += s_part[ps * chunks + chunk_l][8 + k]; }
                } else {
                    gsum += s_part[chunk_l][k]; gsq += s_part[chunk_l][8 + k];
                }
            }
        }
        __syncthreads();
        if (chunks <= (int)blockDim.x) break;
    }
    if ((int)threadIdx.x < a.G) {
        float* dst = a.stats + (((long)n * gridDim.x + blockIdx.x) * a.G + threadIdx.x) * 2;
        dst[0] = gsum; dst[1] = gsq;
    }
}

// Statistics from the producers (round 5): the convolution / projection that wrote src0 (and src1) left, per 32-pixel block and
// channel, the pair {mean, M2 = sum of squared deviations from that mean} of the fp16 values it stored (igemm_device.h,
// gstat_block).  One workgroup per (group, sample) folds the cpg x (H*W / 32) pairs of its group with the parallel-variance
// (Chan) update - every pair carries 32 elements - in a FIXED order: thread t takes pairs t, t + 256, ... sequentially, then a
// binary tree over the 256 threads.  No E[x^2] - mean^2 anywhere, so |mean| >> sigma is harmless, and no second pass over the tensor.
struct GnAcc { float n, mean, m2; };
__device__ __forceinline__ void gn_acc_merge(GnAcc& a, const GnAcc& b) {
    if (b.n == 0.f) return;
    if (a.n == 0.f) { a = b; return; }
    const float n = a.n + b.n, d = b.mean - a.mean;
    a.mean += d * (b.n / n);
    a.m2 += b.m2 + d * d * (a.n * b.n / n);
    a.n = n;
}
__global__ void __launch_bounds__(256)
gn_finalize_kernel(GNArgs a) {
    __shared__ GnAcc s_acc[256];
    const int C = a.C0 + a.C1, cpg = C / a.G;
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int nb = (a.H * a.W) >> 5;                       // 32-pixel blocks per sample (H * W % 32 == 0, checked by the launcher)
    const long P = (long)cpg * nb;
    GnAcc acc = {0.f, 0.f, 0.f};
    for (long i = tid; i < P; i += 256) {
        const int b = (int)(i / cpg), cl = (int)(i - (long)b * cpg);
        const int c = g * cpg + cl;
        const float* gst; int cs, Cs;
        if (c < a.C0) { gst = a.gst0; cs = c; Cs = a.C0; } else { gst = a.gst1; cs = c - a.C0; Cs = a.C1; }
        const float2 pr = *reinterpret_cast<const float2*>(gst + (((long)n * nb + b) * Cs + cs) * 2);
        const GnAcc e = {32.f, pr.x, pr.y};
        gn_acc_merge(acc, e);
    }
    s_acc[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { GnAcc x = s_acc[tid]; gn_acc_merge(x, s_acc[tid + o]); s_acc[tid] = x; }
        __syncthreads();
    }
    if (tid == 0) {
        const GnAcc t = s_acc[0];
        float var = t.m2 / t.n;
        var = var < 0.f ? 0.f : var;
        float* dst = a.stats + ((long)n * a.G + g) * 2;
        dst[0] = t.mean; dst[1] = rsqrtf(var + a.eps);
    }
}

// Pass 2: apply.  Prologue: the fixed-order reduction of the nblk (<= 256) stats blocks to (mean, rstd) per group,
// recomputed by every workgroup (16 .. 64 KB of L2 reads).
__global__ void __launch_bounds__(256)
gn_apply_kernel(GNArgs a) {
    __shared__ float s_mean[64], s_rstd[64];
    const int C = a.C0 + a.C1;
    const int cpg = C / a.G;
    const int chunks = C / 8;
    const int n = blockIdx.y;
    const int HW = a.H * a.W;
    const int p0 = blockIdx.x * a.pix_per_block;
    const int p1 = min(p0 + a.pix_per_block, HW);
    if (a.pre) {                                                     // (mean, rstd) per group were finalised by gn_finalize_kernel
        if ((int)threadIdx.x < a.G) {
            const float2 mr = *reinterpret_cast<const float2*>(a.stats + ((long)n * a.G + threadIdx.x) * 2);
            s_mean[threadIdx.x] = mr.x; s_rstd[threadIdx.x] = mr.y;
        }
    } else {
        // thread (g = t % G, slice = t / G) sums stats blocks slice, slice + NSL, ... (independent loads); thread g then adds
        // the NSL slices in order: a fixed summation order, one round of memory latency
        __shared__ float s_ps[8][64][2];
        const int NSL = 256 / a.G < 8 ? 256 / a.G : 8;               // slices (G = 32: 8)
        const int g = threadIdx.x % a.G, sl = threadIdx.x / a.G;
        if (sl < NSL) {
            float s = 0.f, q = 0.f;
            for (int b = sl; b < a.nblk; b += NSL) {
                const float* p = a.stats + (((long)n * a.nblk + b) * a.G + g) * 2;
                s += p[0]; q += p[1];
            }
            s_ps[sl][g][0] = s; s_ps[sl][g][1] = q;
        }
        __syncthreads();
        if ((int)threadIdx.x < a.G) {
            float s = 0.f, q = 0.f;
            for (int k = 0; k < NSL; ++k) { s += s_ps[k][g][0]; q += s_ps[k][g][1]; }
            const float inv_cnt = 1.0f / ((float)cpg * (float)HW);
            const float dm = s * inv_cnt;                      // mean - pivot
            float var = q * inv_cnt - dm * dm;                 // benign: |dm| is a few sigma at most
            var = var < 0.f ? 0.f : var;
            s_mean[g] = gn_pivot(a, n, g * cpg) + dm; s_rstd[g] = rsqrtf(var + a.eps);
        }
    }
    __syncthreads();
    const int ppi = max(1, (int)blockDim.x / chunks);
    // padded offset of pixel p of sample n without an integer division: pad_off = base + p + 2*(p / W), and
    // p / W = floor((p + 0.5) / W) is exact in fp32 here (p < 2^20, H <= 1024: error 1e-4 << margin 0.5 / W)
    const float invW = 1.0f / (float)a.W;
    const long pbase = (long)n * (a.H + 2) * (a.W + 2) + (a.W + 2) + 1;
    auto poff = [&](int p) -> long { return pbase + p + 2 * (int)(((float)p + 0.5f) * invW); };
    for (int cbase = 0; cbase < chunks; cbase += blockDim.x) {
        int chunk, psub;
        if (chunks <= (int)blockDim.x) { chunk = threadIdx.x % chunks; psub = threadIdx.x / chunks; }
        else { chunk = cbase + threadIdx.x; psub = 0; }
        const bool active = (chunk < chunks) && (psub < ppi);
        if (active) {
            const int c = chunk * 8;
            const half_t* src; int cs, Cs;
            if (c < a.C0) { src = a.src0; cs = c; Cs = a.C0; } else { src = a.src1; cs = c - a.C0; Cs = a.C1; }
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int g = (c + k) / cpg;
                const float ga = a.gamma[c + k] * s_rstd[g];
                sc[k] = ga; sh[k] = a.beta[c + k] - s_mean[g] * ga;
            }
            auto one = [&](int p, const half8_t v, long po) {
                half8_t o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)v[k] * sc[k] + sh[k];
                    if (a.silu) f = silu_f(f);
                    o[k] = (half_t)f;
                }
                const long orow = a.dst_padded ? po : ((long)n * HW + p);
                *reinterpret_cast<half8_t*>(a.dst + orow * C + c) = o;
            };
            int p = p0 + psub;
            for (; p + 3 * ppi < p1; p += 4 * ppi) {       // 4 independent 16-B loads in flight per thread
                half8_t v[4]; long po[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { po[u] = poff(p + u * ppi); v[u] = *reinterpret_cast<const half8_t*>(src + po[u] * Cs + cs); }
#pragma unroll
                for (int u = 0; u < 4; ++u) one(p + u * ppi, v[u], po[u]);
            }
            for (; p < p1; p += ppi) {
                const long po = poff(p);
                one(p, *reinterpret_cast<const half8_t*>(src + po * Cs + cs), po);
            }
        }
        if (chunks <= (int)blockDim.x) break;
    }
}

int g_gn_mode = 0;        // 0 = auto, 1 = always the two-launch form, 2 = slab kernel whenever the slab fits
int g_ln_rpw = 0;         // LayerNorm rows per wave: 0 = by row count, 1 / 2 / 4 forced (diagnostics)

template <int MAXCH, int NT>
void launch_gn_slab(const GNArgs& a, int wgs, hipStream_t s) {
    hipLaunchKernelGGL((gn_slab_kernel<MAXCH, NT>), dim3(wgs), dim3(NT), 0, s, a);
}

// ---- LayerNorm: RPW token rows per wave, rows held in registers ----------------
// One row is only 640 B .. 2.5 KB: with one row per wave the kernel is latency- not bandwidth-bound (a CU has
// 32 waves x 640 B = 20 KB in flight, the HBM pipe wants ~64 KB per CU).  Each wave therefore loads RPW rows before it
// reduces any of them; the per-row arithmetic (8-byte chunks lane + 64 j, butterfly sums, exact two-pass variance) is
// the same for every RPW, so the result does not depend on it.
template <int MAXV, int RPW>   // MAXV = max 16-byte chunks (8 channels) per lane: C <= 64 * 8 * MAXV
__global__ void __launch_bounds__(256)
layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma,
                 const float* __restrict__ beta, long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int chunks = C / 8;
    // ALL loads of the wave's rows are issued before anything waits for one of them: unconditional, with the chunk index
    // clamped (round 2's `if (ch < chunks) { load; use }` put every load in its own block with its own s_waitcnt vmcnt(0): a row
    // of C = 1280 was five dependent memory round trips, and the kernel was latency-bound at 2 TB/s)
    half8_t raw[RPW][MAXV];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < rows ? row0 + r : rows - 1;      // clamp: the tail rows are computed twice, stored once
        const half_t* xr = x + row * C;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int ch = lane + j * 64;
            raw[r][j] = *reinterpret_cast<const half8_t*>(xr + (ch < chunks ? ch : chunks - 1) * 8);
        }
    }
    float v[RPW][MAXV][8];
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) ln_row_stats<MAXV>(raw[r], chunks, C, eps, lane, v[r], mean[r], rstd[r]);      // (common.h)
    // gamma / beta of this lane's chunks: again every load first
    float4 g[MAXV][2], be[MAXV][2];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int ch = lane + j * 64;
        const int cc = (ch < chunks ? ch : chunks - 1) * 8;
        g[j][0] = *reinterpret_cast<const float4*>(gamma + cc); g[j][1] = *reinterpret_cast<const float4*>(gamma + cc + 4);
        be[j][0] = *reinterpret_cast<const float4*>(beta + cc); be[j][1] = *reinterpret_cast<const float4*>(beta + cc + 4);
    }
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int ch = lane + j * 64;
        if (ch < chunks) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (row0 + r < rows) *reinterpret_cast<half8_t*>(y + (row0 + r) * C + ch * 8) = ln_row_affine(v[r][j], mean[r], rstd[r], g[j], be[j]);
        }
    }
}

// ---- row softmax in place (VAE mid-block attention: one 512-wide head, 4096 / 16384 keys) --------
template <int MAXC>   // max 16-B chunks per thread (ncols <= 256*8*MAXC)
__global__ void __launch_bounds__(256)
softmax_rows_kernel(half_t* __restrict__ s, int ncols) {
    __shared__ float red[8];
    half_t* row = s + (long)blockIdx.x * ncols;
    const int nchunks = ncols / 8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float v[MAXC][8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = tid + j * 256;
        if (c < nchunks) {
            const half8_t h = *reinterpret_cast<const half8_t*>(row + c * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[j][k] = (float)h[k] * 1.4426950408889634f; mx = fmaxf(mx, v[j][k]); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = tid + j * 256;
        if (c < nchunks) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[j][k] = __builtin_amdgcn_exp2f(v[j][k] - mx); sum += v[j][k]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) red[4 + wid] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int c = tid + j * 256;
        if (c < nchunks) {
            half8_t o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (half_t)(v[j][k] * inv);
            *reinterpret_cast<half8_t*>(row + c * 8) = o;
        }
    }
}

}  // namespace

extern "C" {

int cfgpp_op_softmax_rows(void* s, long rows, int ncols, void* stream) {
    CFGPP_REQUIRE(s && rows > 0 && ncols % 8 == 0 && ncols <= 256 * 8 * 8, "softmax_rows: ncols=%d (multiple of 8, <= 16384)", ncols);
    hipStream_t st = (hipStream_t)stream;
    const int need = cdiv(ncols / 8, 256);
    if (need <= 2) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3(rows), dim3(256), 0, st, (half_t*)s, ncols);
    else hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3(rows), dim3(256), 0, st, (half_t*)s, ncols);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}


// GroupNorm(+SiLU) over the channel-concat of src0[N,H+2,W+2,C0] and src1[N,H+2,W+2,C1]
// (src1 may be NULL / C1 = 0).  stats: device scratch of N * (1024*G*2 + G*2) floats (the two-launch form uses
// the first N * 64 * G * 2 of them).
void cfgpp_groupnorm_set_mode(int mode) { g_gn_mode = mode; }
static int g_gn_prestats = 1;
void cfgpp_groupnorm_set_prestats(int on) { g_gn_prestats = on ? 1 : 0; }
int cfgpp_groupnorm_prestats_enabled() { return g_gn_prestats; }
void cfgpp_layernorm_set_rows_per_wave(int rpw) { g_ln_rpw = (rpw == 1 || rpw == 2 || rpw == 4) ? rpw : 0; }

int cfgpp_op_groupnorm(const void* src0, const void* src1, void* dst, const float* gamma, const float* beta,
                       float* stats, int N, int H, int W, int C0, int C1, int G, float eps, int silu,
                       int dst_padded, void* stream) {
    const int C = C0 + C1;
    CFGPP_REQUIRE(G > 0 && G <= 64 && C % G == 0, "groupnorm: C=%d not divisible by G=%d (G<=64)", C, G);
    CFGPP_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0, "groupnorm: C0=%d C1=%d must be multiples of 8", C0, C1);
    CFGPP_REQUIRE(src0 && dst && gamma && beta && stats && (C1 == 0 || src1), "groupnorm: null pointer");
    hipStream_t s = (hipStream_t)stream;
    GNArgs a;
    a.src0 = (const half_t*)src0; a.src1 = (const half_t*)src1; a.dst = (half_t*)dst;
    a.gamma = gamma; a.beta = beta;
    a.N = N; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.G = G; a.eps = eps; a.silu = silu;
    a.dst_padded = dst_padded; a.stats = stats; a.nblk = 0; a.gs = 1; a.pix_per_block = 0;
    a.gst0 = nullptr; a.gst1 = nullptr; a.pre = 0;
    const int HW = H * W;
    const int cpg = C / G;
    // ---- (1) slab kernel: smallest gs (groups per workgroup) whose channel range is whole 16-B chunks ----
    if (g_gn_mode != 1 && cpg >= 8) {
        int gs = 0;
        for (int t = 1; t <= 4; t <<= 1) if (G % t == 0 && (t * cpg) % 8 == 0) { gs = t; break; }
        const int cpp = gs ? gs * cpg / 8 : 0;
        // threads = cpp * R, a multiple of 64, at most 960: cpp = 5, 10 -> 320 (x2, x3), 15, 30 -> 960
        int nt = 0;
        if (cpp && cpp <= 32) {
            for (int cand : {320, 640, 960}) {
                if (cand % cpp) continue;
                const int chunks_per_thread = cdiv(HW, cand / cpp);
                if (chunks_per_thread <= 16) { nt = cand; break; }      // 16 chunks = 64 data VGPRs: no spills at 960 threads
            }
        }
        if (nt) {
            const int R = nt / cpp, cpt = cdiv(HW, R);
            const int wgs = N * (G / gs);
            // with few workgroups every CU streams a large slab alone: the two-launch form wins on bandwidth there
            const bool worth = g_gn_mode == 2 || wgs >= 48;
            if (worth) {
                a.gs = gs;
#define GN_SLAB(NT_)                                                                        \
                if (nt == NT_) {                                                            \
                    if (cpt <= 2) launch_gn_slab<2, NT_>(a, wgs, s);                        \
                    else if (cpt <= 4) launch_gn_slab<4, NT_>(a, wgs, s);                   \
                    else if (cpt <= 8) launch_gn_slab<8, NT_>(a, wgs, s);                   \
                    else launch_gn_slab<16, NT_>(a, wgs, s);                                \
                }
                GN_SLAB(320) GN_SLAB(640) GN_SLAB(960)
#undef GN_SLAB
                CFGPP_HIP_CHECK(hipGetLastError());
                return 0;
            }
        }
    }
    // ---- (2) two launches: 64 .. 256 stats blocks per sample (the apply prologue re-reduces them; more of them only
    // when the batch is too small to fill the CUs otherwise: the VAE at batch 1), >= 16 pixels each ----
    const int nblk_max = std::max(64, std::min(256, cdiv(768, N)));
    int ppb = 16;
    while (cdiv(HW, ppb) > nblk_max) ppb <<= 1;
    a.pix_per_block = ppb;
    a.nblk = cdiv(HW, ppb);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a.nblk, N), dim3(256), 0, s, a);
    // apply: aim for >= ~1024 blocks of >= 16 pixels
    GNArgs b = a;
    int apb = 64;
    while (apb > 16 && (long)N * cdiv(HW, apb) < 1024) apb >>= 1;
    b.pix_per_block = apb;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(HW, apb), N), dim3(256), 0, s, b);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

// The same GroupNorm with the statistics taken from the producers of src0 / src1 (gst0 / gst1: [N * H * W / 32][C0 or C1][2] fp32,
// IGemmArgs::gstat): a tiny finalize launch (N x G workgroups) + the apply launch - one read and one write of the tensor at full
// parallelism at every feature-map size, instead of the statistics pass (64x64 level and up) or the one-workgroup-per-slab kernel
// (32x32 and below).  stats: >= N * G * 2 floats.
int cfgpp_op_groupnorm_pre(const void* src0, const void* src1, void* dst, const float* gamma, const float* beta,
                           const float* gst0, const float* gst1, float* stats, int N, int H, int W, int C0, int C1, int G,
                           float eps, int silu, int dst_padded, void* stream) {
    const int C = C0 + C1;
    CFGPP_REQUIRE(G > 0 && G <= 64 && C % G == 0, "groupnorm_pre: C=%d not divisible by G=%d (G<=64)", C, G);
    CFGPP_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0, "groupnorm_pre: C0=%d C1=%d must be multiples of 8", C0, C1);
    CFGPP_REQUIRE((H * W) % 32 == 0, "groupnorm_pre: H*W=%d must be a multiple of 32 (the producers' statistics blocks)", H * W);
    CFGPP_REQUIRE(src0 && dst && gamma && beta && stats && gst0 && (C1 == 0 || (src1 && gst1)), "groupnorm_pre: null pointer");
    hipStream_t s = (hipStream_t)stream;
    GNArgs a;
    a.src0 = (const half_t*)src0; a.src1 = (const half_t*)src1; a.dst = (half_t*)dst;
    a.gamma = gamma; a.beta = beta;
    a.N = N; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.G = G; a.eps = eps; a.silu = silu;
    a.dst_padded = dst_padded; a.stats = stats; a.nblk = 0; a.gs = 1;
    a.gst0 = gst0; a.gst1 = gst1; a.pre = 1;
    const int HW = H * W;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(G, N), dim3(256), 0, s, a);
    int apb = 64;
    while (apb > 16 && (long)N * cdiv(HW, apb) < 1024) apb >>= 1;
    a.pix_per_block = apb;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(cdiv(HW, apb), N), dim3(256), 0, s, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_layernorm(const void* x, void* y, const float* gamma, const float* beta, long rows, int C,
                       float eps, void* stream) {
    CFGPP_REQUIRE(C % 8 == 0 && C <= 64 * 8 * 4, "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    CFGPP_REQUIRE(x && y && gamma && beta && rows > 0, "layernorm: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int need = cdiv(C / 8, 64);
    // rows per wave: 2 for the 640-byte rows of the C = 320 level when there are enough rows to keep the CUs full
    const int rpw = g_ln_rpw > 0 ? g_ln_rpw : (C <= 320 && rows >= 2L * 4096 ? 2 : 1);
#define LN_LAUNCH(MV, RP) hipLaunchKernelGGL((layernorm_kernel<MV, RP>), dim3(cdiv(rows, 4 * RP)), dim3(256), 0, s, \
                                             (const half_t*)x, (half_t*)y, gamma, beta, rows, C, eps)
#define LN_BY_RPW(MV) do { if (rpw >= 4) LN_LAUNCH(MV, 4); else if (rpw == 2) LN_LAUNCH(MV, 2); else LN_LAUNCH(MV, 1); } while (0)
    if (need <= 1) LN_BY_RPW(1);
    else if (need <= 2) LN_BY_RPW(2);
    else if (need <= 3) LN_BY_RPW(3);
    else LN_LAUNCH(4, 1);
#undef LN_BY_RPW
#undef LN_LAUNCH
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
