// Small / boundary kernels of the UNet: conv_in (tiny Cin), conv_out (tiny Cout),
// sinusoidal timestep embedding (K10) and the skinny GEMM used by the time /
// added-condition MLPs and the batched time_emb_proj (M = UNet batch rows <= 64).
// None of them matters for FLOPs; they exist so that the whole forward stays on
// the device, asynchronous, with no host round trip.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// conv_in: 3x3 pad 1, Cin <= 8, NCHW latent (fp32 or fp16) -> halo-padded NHWC fp16.
// Output row r reads latent sample (r % zB): the reference's torch.cat([zt]*2)
// (latent_diffusion.py:153) is an index computation here (K13 eliminated).
// w: [9*Cin][Cout] fp32 (k = tap*Cin + ci), bias [Cout] fp32.
// ---------------------------------------------------------------------------
template <typename TIN>
__global__ void __launch_bounds__(256)
conv_in_kernel(const TIN* __restrict__ z, half_t* __restrict__ out, const float* __restrict__ w,
               const float* __restrict__ bias, int R, int zB, int Cin, int H, int W, int Cout,
               const float* __restrict__ pre_w, const float* __restrict__ pre_b, float in_scale) {
    constexpr int TP = 16;                 // pixels per block
    __shared__ float patch[TP][9 * 8];
    const int r = blockIdx.y;
    const int p0 = blockIdx.x * TP;
    const int HW = H * W;
    const int zb = r % zB;
    const int K = 9 * Cin;
    for (int i = threadIdx.x; i < TP * K; i += blockDim.x) {
        const int tp = i / K, k = i - tp * K;
        const int tap = k / Cin, ci = k - tap * Cin;
        const int p = p0 + tp;
        float v = 0.f;
        if (p < HW) {
            const int y = p / W + tap / 3 - 1, x = p % W + tap % 3 - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                if (pre_w) {
                    // pointwise pre-conv folded in (VAE: z / scaling_factor -> post_quant_conv 1x1), applied to
                    // in-bounds pixels only so the zero padding of the 3x3 conv stays zero
                    float acc = pre_b ? pre_b[ci] : 0.f;
                    for (int j = 0; j < Cin; ++j)
                        acc += pre_w[ci * Cin + j] * (float)(half_t)((float)z[((long)(zb * Cin + j) * H + y) * W + x] * in_scale);
                    v = acc;
                } else {
                    v = (float)z[((long)(zb * Cin + ci) * H + y) * W + x] * in_scale;
                }
            }
        }
        // the reference feeds the UNet an fp16 sample under autocast: round the input once
        patch[tp][k] = (float)(half_t)v;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        float acc[TP];
        const float b = bias ? bias[co] : 0.f;
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) acc[tp] = b;
        for (int k = 0; k < K; ++k) {
            const float wv = w[(long)k * Cout + co];
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) acc[tp] += wv * patch[tp][k];
        }
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
            const int p = p0 + tp;
            if (p < HW) {
                const int y = p / W, x = p - y * W;
                out[((long)(r * (H + 2) + y + 1) * (W + 2) + x + 1) * Cout + co] = (half_t)acc[tp];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// conv_out: 3x3 pad 1, Cout <= 4, halo-padded NHWC fp16 -> NCHW (fp16 or fp32).
// One wave per output pixel, lanes split K = 9*C in 8-channel chunks.
// w: [Cout][9][C] fp16.
// ---------------------------------------------------------------------------
template <int CO, typename TOUT>
__global__ void __launch_bounds__(256)
conv_out_kernel(const half_t* __restrict__ x, TOUT* __restrict__ out, const half_t* __restrict__ w,
                const float* __restrict__ bias, int R, int H, int W, int C, int cout_real,
                float post_scale, float post_shift, int clamp01) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)R * H * W;
    if (pix >= total) return;
    const int HW = H * W;
    const int r = (int)(pix / HW), p = (int)(pix - (long)r * HW);
    const int y = p / W, xq = p - y * W;
    const int cpt = C / 8;                      // chunks per tap
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int c = lane; c < 9 * cpt; c += 64) {
        const int tap = c / cpt, cc = (c - tap * cpt) * 8;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const half8_t xv = *reinterpret_cast<const half8_t*>(
            x + ((long)(r * (H + 2) + y + 1 + dy) * (W + 2) + xq + 1 + dx) * C + cc);
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            const half8_t wv = *reinterpret_cast<const half8_t*>(w + ((long)o * 9 + tap) * C + cc);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[o] += (float)xv[k] * (float)wv[k];
        }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc[o] += __shfl_xor(acc[o], s);
    if (lane < cout_real) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) if (lane == o) v = acc[o];
        v += bias ? bias[lane] : 0.f;
        v = v * post_scale + post_shift;                 // identity (1, 0), or the sampler's `img / 2 + 0.5`
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);      // `.clamp(0, 1)` (latent_diffusion.py:677)
        out[((long)(r * cout_real + lane) * H + y) * W + xq] = (TOUT)v;
    }
}

// ---------------------------------------------------------------------------
// conv_out on large maps (the VAE decoder's last conv: 128 channels -> 3 at 512^2 / 1024^2): the wave-per-pixel kernel
// above costs 1.6 ms per 8 x 512^2 images against a 0.15 ms read-once floor (profiles/r03/vae_b8_64.txt): 2 M waves, each
// with 2.25 exposed load round trips, a 6-step shuffle reduction per output and three scattered 4-byte stores.  Here a
// workgroup owns an 8 x 32 pixel tile (thread = pixel): per 64-channel block the 10 x 34 halo tile goes HBM -> LDS once
// (coalesced 128-byte pixel rows; 144-byte LDS pitch: the 16 lanes of a ds_read_b128 group - consecutive pixels - fall on
// 36 i mod 64, sixteen different bank quads), every thread walks its 9 taps x 8 chunks out of LDS, the weights are wave-uniform
// and come through the scalar cache, two MACs per v_dot2_f32_f16, and a wave stores 2 x 128 contiguous bytes per output plane.
// ---------------------------------------------------------------------------
template <int CO, typename TOUT>
__global__ void __launch_bounds__(256)
conv_out_tile_kernel(const half_t* __restrict__ x, TOUT* __restrict__ out, const half_t* __restrict__ w,
                     const float* __restrict__ bias, int R, int H, int W, int C, int cout_real,
                     float post_scale, float post_shift, int clamp01) {
    constexpr int TH = 8, TW = 32, HH = TH + 2, HW_ = TW + 2, PITCH = 144;
    __shared__ __attribute__((aligned(16))) char tile[HH * HW_ * PITCH];
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int b = blockIdx.x;
    const int txi = b % tiles_x; b /= tiles_x;
    const int tyi = b % tiles_y; const int r = b / tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;                  // first output pixel of the tile = halo-tile origin in PADDED coordinates
    const int ty = tid >> 5, tx = tid & 31;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    const half_t* xr = x + (long)r * (H + 2) * (W + 2) * C;
    for (int cb = 0; cb < C; cb += 64) {
        __syncthreads();                                     // the previous block's reads are done
        // halo tile: HH x HW_ pixels x 8 chunks of 16 bytes; padded coordinates (y0 + hy, x0 + hx) exist for hy <= H + 1 - y0, ...
        for (int i = tid; i < HH * HW_ * 8; i += 256) {
            const int px = i >> 3, ch = i & 7;
            const int hy = px / HW_, hx = px - hy * HW_;
            const int gy = y0 + hy, gx = x0 + hx;
            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (gy < H + 2 && gx < W + 2) v = *reinterpret_cast<const half8_t*>(xr + ((long)gy * (W + 2) + gx) * C + cb + ch * 8);
            *reinterpret_cast<half8_t*>(tile + px * PITCH + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * (tap / 3);
            const char* px = tile + ((ty + dy) * HW_ + tx + dx) * PITCH;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const half8_t xv = *reinterpret_cast<const half8_t*>(px + ch * 16);
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const half8_t wv = *reinterpret_cast<const half8_t*>(w + ((long)o * 9 + tap) * C + cb + ch * 8);      // wave-uniform: scalar loads
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[o] = __builtin_amdgcn_fdot2((h2){xv[2 * k], xv[2 * k + 1]}, (h2){wv[2 * k], wv[2 * k + 1]}, acc[o], false);
                }
            }
        }
    }
    const int y = y0 + ty, xq = x0 + tx;
    if (y < H && xq < W) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            if (o < cout_real) {
                float v = acc[o] + (bias ? bias[o] : 0.f);
                v = v * post_scale + post_shift;
                if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
                out[((long)(r * cout_real + o) * H + y) * W + xq] = (TOUT)v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Sinusoidal embedding (diffusers get_timestep_embedding, flip_sin_to_cos=True,
// downscale_freq_shift=0): out[i][0:half] = cos(v_i * e_j), out[i][half:] = sin(v_i * e_j),
// e_j = exp(-ln(10000) * j / half).  Values are rounded through fp16 like the
// reference's `t_emb.to(dtype=sample.dtype)`.
// vals: device pointer to `count` floats, or null -> use `scalar` for all.
// ---------------------------------------------------------------------------
__global__ void sinusoid_kernel(const float* __restrict__ vals, float scalar, float* __restrict__ out,
                                int count, int dim, int out_ld, int out_off) {
    const int i = blockIdx.x;
    const int half_dim = dim / 2;
    const float v = vals ? vals[i] : scalar;
    for (int j = threadIdx.x; j < half_dim; j += blockDim.x) {
        const float e = expf(-9.210340371976184f * (float)j / (float)half_dim);
        const float arg = v * e;
        out[(long)i * out_ld + out_off + j] = (float)(half_t)cosf(arg);
        out[(long)i * out_ld + out_off + half_dim + j] = (float)(half_t)sinf(arg);
    }
}

// ---------------------------------------------------------------------------
// skinny GEMM: out[m][n] = sum_k act(x[m][k]) * W[n][k] + bias[n] (+ addend[m or 0][n]),
// x fp32 [M][K] (row stride ldx, or stride 0 = broadcast row), W fp16 [N][K], out fp32.
// Each wave: MB rows x NB columns, lanes split K in 8-element chunks.
// ---------------------------------------------------------------------------
template <int MB, int NB>
__global__ void __launch_bounds__(256)
skinny_gemm_kernel(const float* __restrict__ x, int ldx, const half_t* __restrict__ w, const float* __restrict__ bias,
                   const float* __restrict__ addend, int add_ld, float* __restrict__ out, int ldo,
                   int M, int N, int K, int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wid) * NB;
    const int m0 = blockIdx.y * MB;
    if (n0 >= N) return;
    float acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = 0.f;
    for (int kc = lane * 8; kc < K; kc += 64 * 8) {
        float xv[MB][8];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = min(m0 + i, M - 1);
            const float4 a = *reinterpret_cast<const float4*>(x + (long)m * ldx + kc);
            const float4 b = *reinterpret_cast<const float4*>(x + (long)m * ldx + kc + 4);
            xv[i][0] = a.x; xv[i][1] = a.y; xv[i][2] = a.z; xv[i][3] = a.w;
            xv[i][4] = b.x; xv[i][5] = b.y; xv[i][6] = b.z; xv[i][7] = b.w;
            if (silu_in) {
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[i][k] = silu_f(xv[i][k]);
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = min(n0 + j, N - 1);
            const half8_t wv = *reinterpret_cast<const half8_t*>(w + (long)n * K + kc);
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[i][j] += xv[i][k] * (float)wv[k];
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) acc[i][j] += __shfl_xor(acc[i][j], s);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int m = m0 + i;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int n = n0 + j;
                if (n >= N) continue;
                float v = acc[i][j] + (bias ? bias[n] : 0.f);
                if (addend) v += addend[(long)m * add_ld + n];
                if (silu_out) v = silu_f(v);
                out[(long)m * ldo + n] = v;
            }
        }
    }
}

__global__ void f16_to_f32_rows_kernel(const half_t* __restrict__ in, float* __restrict__ out, int rows, int cols,
                                       int out_ld, int out_off) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        out[(long)r * out_ld + out_off + c] = (float)in[i];
    }
}

}  // namespace


// ---------------------------------------------------------------------------
// AutoencoderKL posterior (diffusers DiagonalGaussianDistribution, reached from
// latent_diffusion.py:117-121 / latent_sdxl.py:150-153):  moments = quant_conv(conv_out) (1x1, 8->8),
// mean | logvar = chunk(moments), logvar clamped to [-30, 20], z = (mean + exp(logvar/2) * noise) * scale.
// noise == null gives the posterior mean.  conv_out / quant_conv outputs are rounded to fp16 like the
// reference's fp16 VAE does; the arithmetic itself is fp32.
__global__ void __launch_bounds__(256)
vae_posterior_kernel(const float* __restrict__ co, const float* __restrict__ qw, const float* __restrict__ qb,
                     const float* __restrict__ noise, float* __restrict__ z, float* __restrict__ moments,
                     int B, int HW, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int b = (int)(i / HW), p = (int)(i - (long)b * HW);
    float x[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)(half_t)co[((long)b * 8 + j) * HW + p];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float acc = qb[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += qw[o * 8 + j] * x[j];
        m[o] = (float)(half_t)acc;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float lv = fminf(fmaxf(m[4 + c], -30.f), 20.f);
        const float n = noise ? noise[((long)b * 4 + c) * HW + p] : 0.f;
        z[((long)b * 4 + c) * HW + p] = (m[c] + __expf(0.5f * lv) * n) * scale;
        if (moments) { moments[((long)b * 8 + c) * HW + p] = m[c]; moments[((long)b * 8 + 4 + c) * HW + p] = lv; }
    }
}

extern "C" {

int cfgpp_op_vae_posterior(const float* conv_out, const float* qw, const float* qb, const float* noise, float* z,
                           float* moments, int B, int HW, float scale, void* stream) {
    CFGPP_REQUIRE(conv_out && qw && qb && z && B > 0 && HW > 0, "vae_posterior: bad args");
    hipLaunchKernelGGL(vae_posterior_kernel, dim3(cdiv((long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream, conv_out, qw, qb, noise, z, moments, B, HW, scale);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_conv_in_ex(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                        int R, int zB, int Cin, int H, int W, int Cout, const float* pre_w, const float* pre_b,
                        float in_scale, void* stream) {
    CFGPP_REQUIRE(Cin >= 1 && Cin <= 8, "conv_in: Cin=%d (<= 8)", Cin);
    CFGPP_REQUIRE(z && out && w && R > 0 && zB > 0, "conv_in: bad args");
    dim3 grid(cdiv((long)H * W, 16), R);
    hipStream_t s = (hipStream_t)stream;
    if (z_is_half)
        hipLaunchKernelGGL(conv_in_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)z, (half_t*)out, w, bias, R, zB, Cin, H, W, Cout, pre_w, pre_b, in_scale);
    else
        hipLaunchKernelGGL(conv_in_kernel<float>, grid, dim3(256), 0, s, (const float*)z, (half_t*)out, w, bias, R, zB, Cin, H, W, Cout, pre_w, pre_b, in_scale);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_conv_in(const void* z, int z_is_half, void* out, const float* w, const float* bias,
                     int R, int zB, int Cin, int H, int W, int Cout, void* stream) {
    return cfgpp_op_conv_in_ex(z, z_is_half, out, w, bias, R, zB, Cin, H, W, Cout, nullptr, nullptr, 1.0f, stream);
}

int cfgpp_op_conv_out_ex(const void* x, void* out, int out_is_half, const void* w, const float* bias,
                         int R, int H, int W, int C, int Cout, float post_scale, float post_shift, int clamp01, void* stream);
static int g_conv_out_tiled = 1;      // A/B switch: 0 = always the wave-per-pixel kernel
void cfgpp_conv_out_set_tiled(int on) { g_conv_out_tiled = on ? 1 : 0; }

int cfgpp_op_conv_out(const void* x, void* out, int out_is_half, const void* w, const float* bias,
                      int R, int H, int W, int C, int Cout, void* stream) {
    return cfgpp_op_conv_out_ex(x, out, out_is_half, w, bias, R, H, W, C, Cout, 1.0f, 0.0f, 0, stream);
}

// out = conv3x3(x) * post_scale + post_shift, optionally clamped to [0, 1] (the VAE decoder's conv_out with the
// sampler's image post-processing `(img / 2 + 0.5).clamp(0, 1)` folded in; v * 0.5 is exact, so the result is the
// reference's two separate fp32 ops bit for bit)
int cfgpp_op_conv_out_ex(const void* x, void* out, int out_is_half, const void* w, const float* bias,
                         int R, int H, int W, int C, int Cout, float post_scale, float post_shift, int clamp01, void* stream) {
    CFGPP_REQUIRE(Cout >= 1 && Cout <= 8 && C % 8 == 0, "conv_out: Cout=%d C=%d", Cout, C);
    CFGPP_REQUIRE(x && out && w, "conv_out: null pointer");
    const long total = (long)R * H * W;
    dim3 grid(cdiv(total, 4));
    hipStream_t s = (hipStream_t)stream;
    if (Cout <= 4 && C % 64 == 0 && (long)H * W >= 128 * 128 && g_conv_out_tiled) {      // large maps (VAE decoder): the LDS-tiled kernel
        dim3 tg((unsigned)((long)R * cdiv(H, 8) * cdiv(W, 32)));
        if (out_is_half)
            hipLaunchKernelGGL((conv_out_tile_kernel<4, half_t>), tg, dim3(256), 0, s, (const half_t*)x, (half_t*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
        else if (Cout <= 3)
            hipLaunchKernelGGL((conv_out_tile_kernel<3, float>), tg, dim3(256), 0, s, (const half_t*)x, (float*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
        else
            hipLaunchKernelGGL((conv_out_tile_kernel<4, float>), tg, dim3(256), 0, s, (const half_t*)x, (float*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
        CFGPP_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (Cout > 4) {       // VAE encoder moments (8 channels); weights must hold 8 output rows
        CFGPP_REQUIRE(!out_is_half, "conv_out: 8-channel output is fp32 only");
        hipLaunchKernelGGL((conv_out_kernel<8, float>), grid, dim3(256), 0, s, (const half_t*)x, (float*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
    } else if (out_is_half)
        hipLaunchKernelGGL((conv_out_kernel<4, half_t>), grid, dim3(256), 0, s, (const half_t*)x, (half_t*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
    else
        hipLaunchKernelGGL((conv_out_kernel<4, float>), grid, dim3(256), 0, s, (const half_t*)x, (float*)out, (const half_t*)w, bias, R, H, W, C, Cout, post_scale, post_shift, clamp01);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_sinusoid(const float* vals, float scalar, float* out, int count, int dim, int out_ld, int out_off,
                      void* stream) {
    CFGPP_REQUIRE(out && count > 0 && dim % 2 == 0, "sinusoid: bad args");
    hipLaunchKernelGGL(sinusoid_kernel, dim3(count), dim3(128), 0, (hipStream_t)stream, vals, scalar, out, count, dim, out_ld, out_off);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_skinny_gemm(const float* x, int ldx, const void* w, const float* bias, const float* addend, int add_ld,
                         float* out, int ldo, int M, int N, int K, int silu_in, int silu_out, void* stream) {
    CFGPP_REQUIRE(x && w && out && M > 0 && N > 0 && K % 8 == 0, "skinny_gemm: bad args (K=%d)", K);
    hipStream_t s = (hipStream_t)stream;
    if (M == 1) {
        dim3 grid(cdiv(N, 4 * 2), 1);
        hipLaunchKernelGGL((skinny_gemm_kernel<1, 2>), grid, dim3(256), 0, s, x, ldx, (const half_t*)w, bias, addend, add_ld, out, ldo, M, N, K, silu_in, silu_out);
    } else {
        dim3 grid(cdiv(N, 4 * 2), cdiv(M, 4));
        hipLaunchKernelGGL((skinny_gemm_kernel<4, 2>), grid, dim3(256), 0, s, x, ldx, (const half_t*)w, bias, addend, add_ld, out, ldo, M, N, K, silu_in, silu_out);
    }
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_op_f16_to_f32_rows(const void* in, float* out, int rows, int cols, int out_ld, int out_off, void* stream) {
    hipLaunchKernelGGL(f16_to_f32_rows_kernel, dim3(cdiv((long)rows * cols, 256) > 1024 ? 1024 : cdiv((long)rows * cols, 256)),
                       dim3(256), 0, (hipStream_t)stream, (const half_t*)in, out, rows, cols, out_ld, out_off);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
