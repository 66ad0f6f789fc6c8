// K8 GroupNorm(32)(+SiLU) and K9 LayerNorm for the SD/SDXL UNet (HBM-bound).
//
// GroupNorm works on halo-padded NHWC fp16 and can read its input as the channel
// concatenation of two tensors (the skip-connection `torch.cat([h, skip], 1)` of
// diffusers' up blocks never materialises).  Statistics are fp32 (autocast keeps
// group_norm / layer_norm in fp32); the normalised, affine, optionally SiLU'd
// value is rounded to fp16 once - exactly the value the following fp16 conv /
// linear consumes in the reference.
//
// Three launches, all deterministic (no atomics, fixed summation order -> bit-identical runs):
// (1) per-block partial sums with coalesced 16-B reads, (2) a tiny fixed-order finalize to
// mean / rstd, (3) apply.  Algorithmic bytes per element: 2 (stats read) + 2 (apply read) + 2 (write).
#include "common.h"

namespace {

struct GNArgs {
    const half_t* src0;
    const half_t* src1;   // may be null when C1 == 0
    half_t* dst;
    const float* gamma;   // [C]
    const float* beta;    // [C]
    float* stats;         // [N][nblk][G][2] fp32 per-block partial (sum, sumsq)
    const float* mean_rstd;   // [N][G][2] written by gn_finalize_kernel
    int N, H, W;
    int C0, C1;           // channels of src0 / src1 (C = C0 + C1), both multiples of 8
    int G;                // groups
    float eps;
    int silu;
    int pix_per_block;
    int dst_padded;       // 1: dst is halo-padded NHWC, 0: dst is token-major [N*H*W][C]
};

__device__ __forceinline__ long pad_off(int n, int y, int x, int H, int W) {
    return ((long)(n * (H + 2) + y + 1) * (W + 2) + x + 1);
}

// Pass 1: per-(sample, pixel-block) partial sums of every group, DETERMINISTIC (no atomics): each thread
// sums 8 channels over its pixels, the block combines them through LDS in a fixed order and writes
// part[n][blk][g][2].  Pass 2 (gn_finalize_kernel) reduces the blocks in a fixed order to mean / rstd.
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
            // 4 independent 16-B loads in flight per thread; the per-thread summation order is unchanged
            int p = p0 + psub;
            for (; p + 3 * ppi < p1; p += 4 * ppi) {
                half8_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8_t*>(src + poff(p + u * ppi) * Cs + cs);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { float f = (float)v[u][k]; s[k] += f; q[k] += f * f; }
            }
            for (; p < p1; p += ppi) {
                const half8_t v = *reinterpret_cast<const half8_t*>(src + poff(p) * Cs + cs);
#pragma unroll
                for (int k = 0; k < 8; ++k) { float f = (float)v[k]; s[k] += f; q[k] += f * f; }
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

// Pass 2: fixed-order reduction over the nblk pixel-blocks -> (mean, rstd) per (sample, group).
// One 64-lane wave per (n, g): lane l sums blocks l, l+64, ... in order, then a fixed xor-shuffle tree.
__global__ void __launch_bounds__(64)
gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mr, int nblk, int G, float inv_cnt, float eps) {
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    float s = 0.f, q = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        const float* p = part + (((long)n * nblk + b) * G + g) * 2;
        s += p[0]; q += p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (threadIdx.x == 0) {
        const float mean = s * inv_cnt;
        float var = q * inv_cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        mr[((long)n * G + g) * 2] = mean; mr[((long)n * G + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

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
    for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
        s_mean[g] = a.mean_rstd[((long)n * a.G + g) * 2]; s_rstd[g] = a.mean_rstd[((long)n * a.G + g) * 2 + 1];
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

// ---- LayerNorm: one wave per token row, row held in registers ----------------
template <int MAXV>   // MAXV = max half4 chunks per lane (C <= 64*4*MAXV)
__global__ void __launch_bounds__(256)
layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const float* __restrict__ gamma,
                 const float* __restrict__ beta, long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = C / 4;
    const half_t* xr = x + row * C;
    float v[MAXV][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int ch = lane + j * 64;
        if (ch < chunks) {
            const half4_t h = *reinterpret_cast<const half4_t*>(xr + ch * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[j][k] = (float)h[k]; sum += v[j][k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[j][k] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int ch = lane + j * 64;
        if (ch < chunks) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float d = v[j][k] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq / (float)C + eps);
    half_t* yr = y + row * C;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int ch = lane + j * 64;
        if (ch < chunks) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + ch * 4);
            const float4 b = *reinterpret_cast<const float4*>(beta + ch * 4);
            half4_t o;
            o[0] = (half_t)((v[j][0] - mean) * rstd * g.x + b.x);
            o[1] = (half_t)((v[j][1] - mean) * rstd * g.y + b.y);
            o[2] = (half_t)((v[j][2] - mean) * rstd * g.z + b.z);
            o[3] = (half_t)((v[j][3] - mean) * rstd * g.w + b.w);
            *reinterpret_cast<half4_t*>(yr + ch * 4) = o;
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
// (src1 may be NULL / C1 = 0).  stats: device scratch of N * (1024*G*2 + G*2) floats.
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
    a.dst_padded = dst_padded;
    const int HW = H * W;
    // aim for >= ~1024 blocks, >= 16 pixels per block, and at most 1024 blocks per sample (scratch bound)
    int ppb = 64;
    while (ppb > 16 && (long)N * cdiv(HW, ppb) < 1024) ppb >>= 1;
    while (cdiv(HW, ppb) > 1024) ppb <<= 1;
    a.pix_per_block = ppb;
    const int nblk = cdiv(HW, ppb);
    float* part = stats;
    float* mr = stats + (size_t)N * 1024 * G * 2;
    a.stats = part; a.mean_rstd = mr;
    dim3 grid(nblk, N);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * G), dim3(64), 0, s, part, mr, nblk, G, 1.0f / ((float)(C / G) * (float)HW), eps);
    hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, s, a);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_layernorm(const void* x, void* y, const float* gamma, const float* beta, long rows, int C,
                       float eps, void* stream) {
    CFGPP_REQUIRE(C % 4 == 0 && C <= 64 * 4 * 8, "layernorm: C=%d must be a multiple of 4 and <= 2048", C);
    CFGPP_REQUIRE(x && y && gamma && beta && rows > 0, "layernorm: bad args");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(rows, 4));
    const int need = cdiv(C / 4, 64);
    if (need <= 2)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, (const half_t*)x, (half_t*)y, gamma, beta, rows, C, eps);
    else if (need <= 5)
        hipLaunchKernelGGL(layernorm_kernel<5>, grid, dim3(256), 0, s, (const half_t*)x, (half_t*)y, gamma, beta, rows, C, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, s, (const half_t*)x, (half_t*)y, gamma, beta, rows, C, eps);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
