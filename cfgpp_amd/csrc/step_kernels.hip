// K12: the fused CFG++ sampler step (the reference's own per-step arithmetic).
//
// One elementwise kernel replaces the ~10 torch launches of
//   noise_pred = noise_uc + lam*(noise_c - noise_uc)          latent_diffusion.py:660
//   z0t = (zt - (1-at).sqrt()*noise_pred)/at.sqrt()            latent_diffusion.py:663
//   zt  = at_prev.sqrt()*z0t + (1-at_prev).sqrt()*noise_uc     latent_diffusion.py:666
// and of the inversion / plain-CFG / k-diffusion (Euler, DPM++2M) variants
// (latent_diffusion.py:179-180,283-286,479-490,708-710,855-866,907-908;
//  latent_sdxl.py:317-318,453-456,741-744,904-919,972-973).
//
// Bit-exactness contract: every rounding the reference performs is reproduced
// (fp16 roundings of the eps products, fp32 latent, IEEE division, NO fma
// contraction).  All scalar coefficients arrive as fp32 kernel arguments; the
// host (cfgpp_amd/coeffs.py) computes them with the reference's own 0-dim
// tensor expressions and pre-rounds to fp16 the ones torch would round.
//
// HBM-bound: per element 4 B z in + 2+2 B eps in + 4 B z0t out + 4 B z out = 16 B.
#include "common.h"

namespace {

// RNE to fp16 of a value that has ALREADY been rounded to fp32 (torch computes the fp16 result of an
// op in fp32 and then rounds: two roundings).  The empty asm makes the fp32 value opaque so the compiler
// cannot fuse mul+convert into v_fma_mixlo_f16, which rounds the exact product ONCE and differs on ties.
__device__ __forceinline__ float h_round(float x) {
    asm volatile("" : "+v"(x));
    return (float)(half_t)x;
}

// eps_hat = eps_uc + lam*(eps_c - eps_uc) with fp16 rounding after each op
__device__ __forceinline__ float cfg_mix_h(float uc, float c, float lam) {
    float d = h_round(__fsub_rn(c, uc));
    float e = h_round(__fmul_rn(d, lam));
    return h_round(__fadd_rn(uc, e));
}
__device__ __forceinline__ float cfg_mix_f(float uc, float c, float lam) {
    return __fadd_rn(uc, __fmul_rn(__fsub_rn(c, uc), lam));
}

// `x / c` as the reference's backend evaluates it.  c > 0: IEEE division (torch-CPU, and every tensor / tensor division).
// c < 0: the host passes c = -fl32(1 / divisor) and the kernel MULTIPLIES by -c: torch's GPU `div` with a CPU 0-dim
// scalar (or python number) divisor - which is what `/ at.sqrt()`, `/ sigma.item()`, `/ (2*r)` and `/ (sigma**2+1)**0.5`
// are in the reference - computes `a * (1/b)` with the reciprocal taken once on the host in fp32
// (ATen BinaryDivTrueKernel.cu, `iter.is_cpu_scalar(2)`): up to 1 ulp away from the true quotient.  Every divisor on
// this path is positive, so the sign is free to carry the choice (cfgpp_amd/coeffs.py: `divisor()`).
__device__ __forceinline__ float div_as_ref(float x, float c) {
    return c < 0.f ? __fmul_rn(x, -c) : __fdiv_rn(x, c);
}

template <bool EPS_HALF>
__global__ void __launch_bounds__(256)
ddim_step_kernel(float* __restrict__ z, float* __restrict__ z0t_out,
                 const void* __restrict__ eps_uc_, const void* __restrict__ eps_c_,
                 float lam, float c1, float c2, float c3, float c4,
                 int tweedie_uc, int renoise_uc, long n4, const float* __restrict__ cdev) {
    if (cdev) { c1 = cdev[0]; c2 = cdev[1]; c3 = cdev[2]; c4 = cdev[3]; }      // graph replay: this step's scalars (same fp32 values)
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        float4 zv = reinterpret_cast<const float4*>(z)[i];
        float uc[4], cc[4];
        if (EPS_HALF) {
            half4_t a = reinterpret_cast<const half4_t*>(eps_uc_)[i];
            half4_t b = reinterpret_cast<const half4_t*>(eps_c_)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) { uc[k] = (float)a[k]; cc[k] = (float)b[k]; }
        } else {
            float4 a = reinterpret_cast<const float4*>(eps_uc_)[i];
            float4 b = reinterpret_cast<const float4*>(eps_c_)[i];
            uc[0] = a.x; uc[1] = a.y; uc[2] = a.z; uc[3] = a.w;
            cc[0] = b.x; cc[1] = b.y; cc[2] = b.z; cc[3] = b.w;
        }
        float zi[4] = {zv.x, zv.y, zv.z, zv.w};
        float z0[4], zn[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float hat = EPS_HALF ? cfg_mix_h(uc[k], cc[k], lam) : cfg_mix_f(uc[k], cc[k], lam);
            float A = tweedie_uc ? uc[k] : hat;
            float B = renoise_uc ? uc[k] : hat;
            float pa = __fmul_rn(A, c1);
            float pb = __fmul_rn(B, c4);
            if (EPS_HALF) { pa = h_round(pa); pb = h_round(pb); }
            z0[k] = div_as_ref(__fsub_rn(zi[k], pa), c2);
            zn[k] = __fadd_rn(__fmul_rn(c3, z0[k]), pb);
        }
        reinterpret_cast<float4*>(z0t_out)[i] = make_float4(z0[0], z0[1], z0[2], z0[3]);
        reinterpret_cast<float4*>(z)[i] = make_float4(zn[0], zn[1], zn[2], zn[3]);
    }
}

// fp16 LATENT variant: the inversion / edit paths start from `vae.encode(...)`, which is fp16 under the
// reference's fp16 pipeline, so zt stays fp16 through the inversion AND the regeneration loop and every op
// rounds to fp16 (latent_diffusion.py:168-180,527-541; latent_sdxl.py:307-318,989-1011):
//     pa = h(c1*A); z0t = h(h(z - pa)/c2); z' = h(h(c3*z0t) + h(c4*B))
__global__ void __launch_bounds__(256)
ddim_step_h_kernel(half_t* __restrict__ z, half_t* __restrict__ z0t_out,
                   const half_t* __restrict__ eps_uc, const half_t* __restrict__ eps_c,
                   float lam, float c1, float c2, float c3, float c4,
                   int tweedie_uc, int renoise_uc, long n4, const float* __restrict__ cdev) {
    if (cdev) { c1 = cdev[0]; c2 = cdev[1]; c3 = cdev[2]; c4 = cdev[3]; }
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        const half4_t zv = reinterpret_cast<const half4_t*>(z)[i];
        const half4_t a = reinterpret_cast<const half4_t*>(eps_uc)[i];
        const half4_t b = reinterpret_cast<const half4_t*>(eps_c)[i];
        half4_t z0, zn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float uc = (float)a[k], cc = (float)b[k];
            const float hat = cfg_mix_h(uc, cc, lam);
            const float A = tweedie_uc ? uc : hat;
            const float B = renoise_uc ? uc : hat;
            const float pa = h_round(__fmul_rn(A, c1));
            const float pb = h_round(__fmul_rn(B, c4));
            const float z0f = h_round(div_as_ref(h_round(__fsub_rn((float)zv[k], pa)), c2));
            const float znf = h_round(__fadd_rn(h_round(__fmul_rn(c3, z0f)), pb));
            z0[k] = (half_t)z0f; zn[k] = (half_t)znf;
        }
        reinterpret_cast<half4_t*>(z0t_out)[i] = z0;
        reinterpret_cast<half4_t*>(z)[i] = zn;
    }
}

// scale the k-diffusion latent into the UNet input: xc = x / s (SD1.5, mode 0; s < 0: div_as_ref) or x * s (SDXL 2M, mode 1)
__global__ void __launch_bounds__(256)
kdiff_input_kernel(const half_t* __restrict__ x, half_t* __restrict__ xc, float s, int mode, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = (float)x[i];
        v = mode == 0 ? div_as_ref(v, s) : __fmul_rn(v, s);
        xc[i] = (half_t)h_round(v);
    }
}

struct KdiffCoef {
    float lam;
    float sigma;        // denoised (SD form):  x - h(eps*sigma)
    float c_out_h;      // denoised (XL form):  x + h(eps*c_out_h), c_out_h = fp16-rounded(-sigma)
    float sigma_item;   // to_d: (x - d_from)/sigma_item   (< 0: -1/sigma, see div_as_ref)
    float sigma_next;   // euler: den + d*sigma_next
    float neg_exp_mh_h; // fp16-rounded(-exp(-h))
    float expm1_mh_h;   // fp16-rounded(expm1(-h))
    float two_r;        // 2*r (fp32; < 0: -1/(2r), see div_as_ref)
    float exp_mh_h;     // fp16-rounded(exp(-h))
};

// variant: 0 = CFG (d_from = den, lead = den, diff = den - old, new_old = den)
//          1 = CFG++ SD1.5 (d_from = uden, lead = uden, diff = den - old,  new_old = uden)
//          2 = CFG++ SDXL  (d_from = uden, lead = uden, diff = uden - old, new_old = uden)
// euler_branch: 1 -> x' = den + ((x - d_from)/sigma)*sigma_next  (also every Euler solver step)
// All arithmetic on fp16 values, each op rounded to fp16 (computed exactly in fp32 first).
__global__ void __launch_bounds__(256)
kdiff_step_kernel(half_t* __restrict__ x, half_t* __restrict__ den_out, half_t* __restrict__ old,
                  const half_t* __restrict__ eps_uc, const half_t* __restrict__ eps_c,
                  KdiffCoef k, int variant, int xl_form, int euler_branch, int write_old, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float xv = (float)x[i];
        const float uc = (float)eps_uc[i], cc = (float)eps_c[i];
        const float hat = cfg_mix_h(uc, cc, k.lam);
        float den, uden;
        if (xl_form) {
            den = h_round(__fadd_rn(xv, h_round(__fmul_rn(hat, k.c_out_h))));
            uden = h_round(__fadd_rn(xv, h_round(__fmul_rn(uc, k.c_out_h))));
        } else {
            den = h_round(__fsub_rn(xv, h_round(__fmul_rn(hat, k.sigma))));
            uden = h_round(__fsub_rn(xv, h_round(__fmul_rn(uc, k.sigma))));
        }
        const float d_from = variant == 0 ? den : uden;
        float xn;
        if (euler_branch) {
            float d = h_round(div_as_ref(h_round(__fsub_rn(xv, d_from)), k.sigma_item));
            xn = h_round(__fadd_rn(den, h_round(__fmul_rn(d, k.sigma_next))));
        } else {
            const float ov = (float)old[i];
            const float lead = d_from;
            const float diff_a = variant == 2 ? uden : den;
            float term1 = h_round(__fmul_rn(lead, k.neg_exp_mh_h));
            float t2 = h_round(__fmul_rn(h_round(__fsub_rn(diff_a, ov)), k.expm1_mh_h));
            t2 = h_round(div_as_ref(t2, k.two_r));
            float extra1 = h_round(__fsub_rn(term1, t2));
            float extra2 = h_round(__fmul_rn(xv, k.exp_mh_h));
            xn = h_round(__fadd_rn(h_round(__fadd_rn(den, extra1)), extra2));
        }
        den_out[i] = (half_t)den;
        if (write_old) old[i] = (half_t)(variant == 0 ? den : uden);
        x[i] = (half_t)xn;
    }
}

// ---- building blocks of the 2-stage / ancestral k-diffusion samplers (fp16 latents) --------------
// den = x - h(eps_hat*sigma), uden = x - h(eps_uc*sigma)           (latent_diffusion.py:235-241)
__global__ void __launch_bounds__(256)
kdiff_denoise_kernel(const half_t* __restrict__ x, const half_t* __restrict__ eps_uc, const half_t* __restrict__ eps_c,
                     float lam, float sigma, half_t* __restrict__ den, half_t* __restrict__ uden, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float xv = (float)x[i], uc = (float)eps_uc[i], cc = (float)eps_c[i];
        const float hat = cfg_mix_h(uc, cc, lam);
        den[i] = (half_t)h_round(__fsub_rn(xv, h_round(__fmul_rn(hat, sigma))));
        uden[i] = (half_t)h_round(__fsub_rn(xv, h_round(__fmul_rn(uc, sigma))));
    }
}
// mode 0: out = h( h(x*a) - h(y*b) )                 x_2 / x of DPM-Solver++(2S)   (latent_diffusion.py:428,435,804)
// mode 1: out = h( h(y - h(z*b)) + h(x*a) )          CFG++ 2S final update          (latent_diffusion.py:811)
// mode 2: out = h( x + h(y*a) )                      ancestral noise                (latent_diffusion.py:379,438)
__global__ void __launch_bounds__(256)
lincomb_kernel(half_t* out, const half_t* x, const half_t* __restrict__ y,
               const half_t* __restrict__ z, float a, float b, int mode, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float xv = (float)x[i], yv = (float)y[i];
        float r;
        if (mode == 0) r = __fsub_rn(h_round(__fmul_rn(xv, a)), h_round(__fmul_rn(yv, b)));
        else if (mode == 1) r = __fadd_rn(h_round(__fsub_rn(yv, h_round(__fmul_rn((float)z[i], b)))), h_round(__fmul_rn(xv, a)));
        else r = __fadd_rn(xv, h_round(__fmul_rn(yv, a)));
        out[i] = (half_t)h_round(r);
    }
}

inline int grid_for(long items) {
    long g = (items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" {

int cfgpp_step_ddim(void* z, void* z0t_out, const void* eps_uc, const void* eps_c, int eps_is_half,
                    float lam, float c1, float c2, float c3, float c4,
                    int tweedie_uc, int renoise_uc, long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && (n % 4) == 0, "cfgpp_step_ddim: n=%ld must be a positive multiple of 4", n);
    CFGPP_REQUIRE(z && z0t_out && eps_uc && eps_c, "cfgpp_step_ddim: null pointer");
    const long n4 = n / 4;
    hipStream_t s = (hipStream_t)stream;
    if (eps_is_half)
        hipLaunchKernelGGL(ddim_step_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, s, (float*)z, (float*)z0t_out,
                           eps_uc, eps_c, lam, c1, c2, c3, c4, tweedie_uc, renoise_uc, n4, (const float*)nullptr);
    else
        hipLaunchKernelGGL(ddim_step_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, s, (float*)z, (float*)z0t_out,
                           eps_uc, eps_c, lam, c1, c2, c3, c4, tweedie_uc, renoise_uc, n4, (const float*)nullptr);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_step_ddim_h(void* z, void* z0t_out, const void* eps_uc, const void* eps_c,
                      float lam, float c1, float c2, float c3, float c4,
                      int tweedie_uc, int renoise_uc, long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && (n % 4) == 0, "cfgpp_step_ddim_h: n=%ld must be a positive multiple of 4", n);
    CFGPP_REQUIRE(z && z0t_out && eps_uc && eps_c, "cfgpp_step_ddim_h: null pointer");
    const long n4 = n / 4;
    hipLaunchKernelGGL(ddim_step_h_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (half_t*)z, (half_t*)z0t_out,
                       (const half_t*)eps_uc, (const half_t*)eps_c, lam, c1, c2, c3, c4, tweedie_uc, renoise_uc, n4, (const float*)nullptr);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- whole-step graph replay (unet.hip: cfgpp_sample_graph_ddim) ------------------------------------------------------------
// A captured step cannot take its scalars as kernel arguments (they are baked at capture), so the graph's first node copies row
// *idx of the per-step table [n][8] = {t, c1, c2, c3, c4, -, -, -} into `cur` and advances *idx; the timestep sinusoid reads
// cur[0], the step kernel cur[1..4].  Same fp32 values as the eager path's arguments -> bit-identical latents.
__global__ void step_advance_kernel(const float* __restrict__ tab, int* __restrict__ idx, float* __restrict__ cur) {
    const int i = *idx;
    if (threadIdx.x < 8) cur[threadIdx.x] = tab[(long)i * 8 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) *idx = i + 1;
}
int step_advance_launch(const float* tab, int* idx, float* cur, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, tab, idx, cur);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}
// the generalised DDIM update with c1..c4 read from cdev[0..3] (device memory)
int step_ddim_dev_launch(void* z, void* z0t_out, const void* eps_uc, const void* eps_c, int eps_is_half, int z_is_half, float lam,
                         const float* cdev, int tweedie_uc, int renoise_uc, long n, hipStream_t s) {
    CFGPP_REQUIRE(n > 0 && (n % 4) == 0 && z && z0t_out && eps_uc && eps_c && cdev, "step_ddim_dev: bad args");
    CFGPP_REQUIRE(!z_is_half || eps_is_half, "step_ddim_dev: an fp16 latent needs fp16 eps");
    const long n4 = n / 4;
    if (z_is_half)
        hipLaunchKernelGGL(ddim_step_h_kernel, dim3(grid_for(n4)), dim3(256), 0, s, (half_t*)z, (half_t*)z0t_out, (const half_t*)eps_uc,
                           (const half_t*)eps_c, lam, 0.f, 1.f, 0.f, 0.f, tweedie_uc, renoise_uc, n4, cdev);
    else if (eps_is_half)
        hipLaunchKernelGGL(ddim_step_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, s, (float*)z, (float*)z0t_out, eps_uc, eps_c,
                           lam, 0.f, 1.f, 0.f, 0.f, tweedie_uc, renoise_uc, n4, cdev);
    else
        hipLaunchKernelGGL(ddim_step_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, s, (float*)z, (float*)z0t_out, eps_uc, eps_c,
                           lam, 0.f, 1.f, 0.f, 0.f, tweedie_uc, renoise_uc, n4, cdev);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" {

int cfgpp_kdiff_input(const void* x, void* xc, float s, int mode, long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && x && xc, "cfgpp_kdiff_input: bad args");
    hipLaunchKernelGGL(kdiff_input_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)xc, s, mode, n);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

// coef[9] = {lam, sigma, c_out_h, sigma_item, sigma_next, neg_exp_mh_h, expm1_mh_h, two_r, exp_mh_h}
int cfgpp_step_kdiff(void* x, void* den_out, void* old, const void* eps_uc, const void* eps_c,
                     const float* coef, int variant, int xl_form, int euler_branch, int write_old,
                     long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && x && den_out && eps_uc && eps_c && coef, "cfgpp_step_kdiff: bad args");
    CFGPP_REQUIRE(euler_branch || old, "cfgpp_step_kdiff: 2M branch needs old_denoised");
    CFGPP_REQUIRE(!write_old || old, "cfgpp_step_kdiff: write_old needs old buffer");
    KdiffCoef k{coef[0], coef[1], coef[2], coef[3], coef[4], coef[5], coef[6], coef[7], coef[8]};
    hipLaunchKernelGGL(kdiff_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       (half_t*)x, (half_t*)den_out, (half_t*)old, (const half_t*)eps_uc, (const half_t*)eps_c,
                       k, variant, xl_form, euler_branch, write_old, n);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_kdiff_denoise(const void* x, const void* eps_uc, const void* eps_c, float lam, float sigma,
                        void* den, void* uden, long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && x && eps_uc && eps_c && den && uden, "cfgpp_kdiff_denoise: bad args");
    hipLaunchKernelGGL(kdiff_denoise_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const half_t*)x,
                       (const half_t*)eps_uc, (const half_t*)eps_c, lam, sigma, (half_t*)den, (half_t*)uden, n);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

int cfgpp_lincomb(void* out, const void* x, const void* y, const void* z, float a, float b, int mode, long n, void* stream) {
    CFGPP_REQUIRE(n > 0 && out && x && y && mode >= 0 && mode <= 2 && (mode != 1 || z), "cfgpp_lincomb: bad args");
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (half_t*)out, (const half_t*)x,
                       (const half_t*)y, (const half_t*)z, a, b, mode, n);
    CFGPP_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
