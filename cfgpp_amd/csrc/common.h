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
