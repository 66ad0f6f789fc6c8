// K8 GroupNorm(32)(+SiLU) and K9 LayerNorm for the SD/SDXL UNet (HBM-bound).
//
// GroupNorm works on halo-padded NHWC fp16 and can read its input as the channel
// concatenation of two tensors (the skip-connection `torch.cat([h, skip], 1)` of
// diffusers' up blocks never materialises).  Statistics are fp32 (autocast keeps
// group_norm / layer_norm in fp32); the normalised, affine, optionally SiLU'd
// value is rounded to fp16 once - exactly the value the following fp16 conv /
// linear consumes in the reference.
//
// Two launches: (1) partial sums per (n, group) with coalesced 16-B reads,
// LDS-atomics per block, one global atomic per (block, group);
// (2) apply.  Algorithmic bytes per element: 2 (stats read) + 2 (apply read) + 2 (write).
#include "common.h"

namespace {

struct GNArgs {
    const half_t* src0;
    const half_t* src1;   // may be null when C1 == 0
    half_t* dst;
    const float* gamma;   // [C]
    const float* beta;    // [C]
    float* stats;         // [N][G][2] fp32 (sum, sumsq), zeroed before the stats kernel
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

__global__ void __launch_bounds__(256)
gn_stats_kernel(GNArgs a) {
    __shared__ float s_acc[2 * 64];   // up to 64 groups
    const int C = a.C0 + a.C1;
    const int cpg = C / a.G;
    const int chunks = C / 8;
    const int n = blockIdx.y;
    const int HW = a.H * a.W;
    const int p0 = blockIdx.x * a.pix_per_block;
    const int p1 = min(p0 + a.pix_per_block, HW);
    for (int i = threadIdx.x; i < 2 * a.G; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
    const int ppi = max(1, (int)blockDim.x / chunks);        // pixels per iteration
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
            for (int p = p0 + psub; p < p1; p += ppi) {
                const int y = p / a.W, x = p - y * a.W;
                const half8_t v = *reinterpret_cast<const half8_t*>(src + pad_off(n, y, x, a.H, a.W) * Cs + cs);
#pragma unroll
                for (int k = 0; k < 8; ++k) { float f = (float)v[k]; s[k] += f; q[k] += f * f; }
            }
            // flush the 8 channels into their groups
            int g_cur = c / cpg; float gs = 0.f, gq = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int g = (c + k) / cpg;
                if (g != g_cur) { atomicAdd(&s_acc[2 * g_cur], gs); atomicAdd(&s_acc[2 * g_cur + 1], gq); g_cur = g; gs = 0.f; gq = 0.f; }
                gs += s[k]; gq += q[k];
            }
            atomicAdd(&s_acc[2 * g_cur], gs); atomicAdd(&s_acc[2 * g_cur + 1], gq);
        }
        if (chunks <= (int)blockDim.x) break;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.G; i += blockDim.x) atomicAdd(&a.stats[(long)n * a.G * 2 + i], s_acc[i]);
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
    const float inv_cnt = 1.0f / ((float)cpg * (float)HW);
    for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
        const float sum = a.stats[((long)n * a.G + g) * 2], sq = a.stats[((long)n * a.G + g) * 2 + 1];
        const float mean = sum * inv_cnt;
        float var = sq * inv_cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        s_mean[g] = mean; s_rstd[g] = rsqrtf(var + a.eps);
    }
    __syncthreads();
    const int ppi = max(1, (int)blockDim.x / chunks);
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
            for (int p = p0 + psub; p < p1; p += ppi) {
                const int y = p / a.W, x = p - y * a.W;
                const long po = pad_off(n, y, x, a.H, a.W);
                const half8_t v = *reinterpret_cast<const half8_t*>(src + po * Cs + cs);
                half8_t o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f = (float)v[k] * sc[k] + sh[k];
                    if (a.silu) f = silu_f(f);
                    o[k] = (half_t)f;
                }
                const long orow = a.dst_padded ? po : ((long)n * HW + p);
                *reinterpret_cast<half8_t*>(a.dst + orow * C + c) = o;
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

}  // namespace

extern "C" {

// GroupNorm(+SiLU) over the channel-concat of src0[N,H+2,W+2,C0] and src1[N,H+2,W+2,C1]
// (src1 may be NULL / C1 = 0).  stats: device scratch of N*G*2 floats.
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
    a.gamma = gamma; a.beta = beta; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.C0 = C0; a.C1 = C1; a.G = G; a.eps = eps; a.silu = silu;
    a.dst_padded = dst_padded;
    const int HW = H * W;
    // aim for >= ~1024 blocks while keeping >= 16 pixels per thread-column
    int ppb = 64;
    while (ppb > 16 && (long)N * cdiv(HW, ppb) < 1024) ppb >>= 1;
    a.pix_per_block = ppb;
    CFGPP_HIP_CHECK(hipMemsetAsync(stats, 0, sizeof(float) * 2 * (size_t)N * G, s));
    dim3 grid(cdiv(HW, ppb), N);
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, s, a);
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
