// Common types and helpers for the gfx950 kernels of libcfgpp_hip.so.
// CDNA4 only: wave64, MFMA f16 32x32x16, 160 KiB LDS.  No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CFGPP_WAVE 64

// ---- error plumbing (no exception crosses the C ABI) -----------------------
void cfgpp_set_error(const char* fmt, ...);
int cfgpp_claim_device(int device_id);      // 0, or -1 (+ error) when the process already drives another device
#define CFGPP_HIP_CHECK(expr)                                                         \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            cfgpp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                            hipGetErrorString(_e));                                   \
            return -1;                                                                \
        }                                                                             \
    } while (0)
#define CFGPP_REQUIRE(cond, ...)                                                      \
    do {                                                                              \
        if (!(cond)) {                                                                \
            cfgpp_set_error(__VA_ARGS__);                                             \
            return -2;                                                                \
        }                                                                             \
    } while (0)

// whole-step graph replay helpers (step_kernels.hip; used by unet.hip: cfgpp_sample_graph_ddim)
int step_advance_launch(const float* tab, int* idx, float* cur, hipStream_t s);
int step_ddim_dev_launch(void* z, void* z0t_out, const void* eps_uc, const void* eps_c, int eps_is_half, int z_is_half, float lam,
                         const float* cdev, int tweedie_uc, int renoise_uc, long n, hipStream_t s);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- activation layouts -----------------------------------------------------
// Spatial activations live in HBM as halo-padded NHWC fp16: [N][H+2][W+2][C],
// halo == 0 for ever (buffers are keyed by shape and only interiors are
// written), so a 3x3 tap gather never needs a bounds check.
// Transformer activations are token-major [N*H*W][C].
struct RowMap {
    // row m of a GEMM operand -> element offset.
    // mode 0: linear            off = m * ld
    // mode 1: padded NHWC       m=(n,y,x) over (H,W): ((n*(H+2)+y+1)*(W+2)+x+1)*ld
    // mode 2: padded, stride 2  source is (2H,2W):   ((n*(2H+2)+2y+1)*(2W+2)+2x+1)*ld
    // mode 3: padded, nearest-2x upsample: source is (H/2,W/2), per-tap address
    int mode;
    int H, W;     // OUTPUT spatial size the rows enumerate (modes 1..3)
    int ld;       // elements per row / pixel
};

// Column of key `tok` in a V^T matrix (attention contract, attn_kernel.hip): within every 32-key block bits 2 and 3
// of the key index are swapped, which makes the 8 keys a lane feeds to one PV MFMA contiguous.
__host__ __device__ __forceinline__ int cfgpp_vt_pos(int tok) {
    return (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-GELU x*Phi(x) with erfc by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the fp16
// output resolution): 1 rcp + 1 exp2 + a degree-5 Horner instead of libm erff (~3x fewer VALU ops in the
// GEGLU epilogue, which evaluates it 4C times per token).
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erfc_ax = poly * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float cdf = x >= 0.f ? 1.0f - 0.5f * erfc_ax : 0.5f * erfc_ax;
    return x * cdf;
}

// The same function for a PAIR of values on the packed-fp32 pipe, without transcendentals.  The GEGLU epilogue evaluates
// GELU 4C times per token: on the 256 x 256 tile that is 64 evaluations per lane, and gelu_erf_f's rcp + exp2 (quarter-rate
// instructions) + 14 VALU operations made the epilogue ~10 900 cycles per workgroup, as long as 3.5 K-tiles of MFMAs
// (profiles/r03/timelines: geglu HW=4096, epilogue 5.5 of 15.2 us).  Here
//      x Phi(x) = x / 2 + |x| * h(min(|x|, c)),      h(a) = erf(a / sqrt 2) / 2,
// with h a degree-12 polynomial in the centred variable t = 2 a / c - 1 (Chebyshev fit converted to monomials: coefficients
// <= 0.5 in magnitude, so fp32 Horner loses nothing), scaled so that h(c) = 1/2 to the last bit of an fp32 Horner evaluation: for
// x <= -c the result is 0 to within 1e-6, for x >= c it is x.  c = 4.75.  Max abs error against the erf form 6.5e-6 over all x
// (tests/test_index_math_cpu.py evaluates THESE constants in numpy fp32), i.e. below the fp16 resolution of the output for every
// |y| >= 0.0133 and below 1e-5 absolute elsewhere.  18 issue slots per pair (2 v_min, 13 v_pk_fma, 2 v_mul, 1 v_pk_fma).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CFGPP_GELU_C 4.75f
#define CFGPP_GELU_POLY {4.912262559e-01f, 5.645360425e-02f, -1.591939032e-01f, 2.464362979e-01f, -1.982378513e-01f, 1.363233291e-02f, \
                         1.386516690e-01f, -1.158297956e-01f, -9.092462249e-03f, 6.090912968e-02f, -1.936562732e-02f, -1.160200126e-02f, \
                         6.012340542e-03f}
__device__ __forceinline__ f32x2 gelu_erf_pk(f32x2 x) {
    constexpr float k[13] = CFGPP_GELU_POLY;
    f32x2 a;
    a.x = fminf(fabsf(x.x), CFGPP_GELU_C); a.y = fminf(fabsf(x.y), CFGPP_GELU_C);
    const f32x2 t = __builtin_elementwise_fma(a, (f32x2){2.0f / CFGPP_GELU_C, 2.0f / CFGPP_GELU_C}, (f32x2){-1.0f, -1.0f});
    f32x2 p = {k[12], k[12]};
#pragma unroll
    for (int i = 11; i >= 0; --i) p = __builtin_elementwise_fma(p, t, (f32x2){k[i], k[i]});
    f32x2 m;
    m.x = fabsf(x.x) * p.x; m.y = fabsf(x.y) * p.y;
    return __builtin_elementwise_fma(x, (f32x2){0.5f, 0.5f}, m);
}

// ---- LayerNorm of ONE row held by a wave (layernorm_kernel): lane l holds the 16-byte chunks l, l + 64, ... (8 channels each) of
// the row; exact two-pass variance.
template <int MAXV>
__device__ __forceinline__ void ln_row_stats(const half8_t (&raw)[MAXV], int chunks, int C, float eps, int lane,
                                             float (&v)[MAXV][8], float& mean, float& rstd) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const bool live = lane + j * 64 < chunks;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[j][k] = live ? (float)raw[j][k] : 0.f; sum += v[j][k]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const bool live = lane + j * 64 < chunks;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[j][k] - mean; sq += live ? d * d : 0.f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    rstd = rsqrtf(sq / (float)C + eps);
}
// chunk j of the normalised row: (v - mean) * rstd * gamma + beta, rounded to fp16
__device__ __forceinline__ half8_t ln_row_affine(const float (&v)[8], float mean, float rstd, const float4 (&g)[2], const float4 (&be)[2]) {
    const float gg[8] = {g[0].x, g[0].y, g[0].z, g[0].w, g[1].x, g[1].y, g[1].z, g[1].w};
    const float bb[8] = {be[0].x, be[0].y, be[0].z, be[0].w, be[1].x, be[1].y, be[1].z, be[1].w};
    half8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (half_t)((v[k] - mean) * rstd * gg[k] + bb[k]);
    return o;
}
