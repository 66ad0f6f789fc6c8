// How far can one SIMD of gfx950 overlap the matrix pipe with the softmax VALU work of the attention kernel?
// (attn64_kernel's key-tile step is 16 x v_mfma_f32_32x32x16_f16 = 512 matrix cycles beside 32 v_exp_f32 +
// 16 packs + 16 max3 of VALU per wave; its measured matrix-pipe utilisation is 0.46-0.48 at four waves per SIMD.)
// One workgroup of 4 x W waves = W waves on every SIMD of one CU; every wave runs `iters` steps of one of these bodies
// between two s_memtime reads, no LDS, no barrier, no memory:
//   mfma     16 MFMAs (two accumulators, chains of 4 like QK^T and PV)
//   valu     32 v_exp_f32 + 16 v_cvt_pkrtz + 16 v_max3_f32
//   chain    the attention dependency: 8 MFMAs -> max3 over their 32 results -> 32 exps of them -> 16 packs -> 8 MFMAs that
//            take the packs as their B operand.  Overlap can only come from OTHER waves of the SIMD.
//   weave    the same instruction mix, but the VALU work is independent of the MFMAs it sits between (what a perfectly
//            software-pipelined loop would present to the issue logic): 2 exp + 1 pack + 1 max3 after every MFMA.
//   pipe     a software-pipelined step with the real dependencies: the 8 QK^T MFMAs of tile j+1 woven with the softmax
//            VALU of tile j (4 exp + 2 packs + 2 max3 after each), then the 8 PV MFMAs of tile j.  Needs two S tiles in
//            registers (about 150 VGPRs), so at most two waves per SIMD.
// Prints cycles per step and the matrix-pipe utilisation W x 512 / cycles for W = 1, 2, 3, 4.
//   hipcc --offload-arch=gfx950 -O3 -o micro_overlap.bin micro_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float exp2_raw(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ unsigned pack_raw(float a, float b) { unsigned r; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

union Frag { h8 h; unsigned u[4]; };

template <int MODE>
__global__ void __launch_bounds__(MODE == 4 ? 512 : 1024) k(unsigned long long* cyc, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    Frag q[2], kf[2], p[4];                                        // (operand values do not matter to the issue logic: two of each)
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) { q[j & 1].h[e] = (_Float16)(0.01f * ((lane + e + j) & 7)); kf[j & 1].h[e] = (_Float16)(0.02f * ((lane * 3 + e + j) & 7)); p[j].h[e] = (_Float16)0.5f; }
    f16v s0, s1, o0, o1, negm, t0v, t1v;                                     // negm: the accumulator seed of S (the kernel's -m broadcast)
    for (int e = 0; e < 16; ++e) { s0[e] = -1.f; s1[e] = -2.f; o0[e] = 0.f; o1[e] = 0.f; negm[e] = -1.f - 0.001f * lane; }
    asm volatile("" : "+v"(negm));
    t0v = s0; t1v = s1;
    constexpr int NX = (MODE == 1 || MODE == 3) ? 32 : 1;          // the free-standing softmax operands exist only where they are used
    float x[NX], m = -3.f;
    for (int e = 0; e < NX; ++e) x[e] = -0.01f * (lane + e);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(negm), "+v"(q[0].h), "+v"(q[1].h));   // nothing in the body is loop-invariant to the compiler
        if constexpr (MODE == 0 || MODE == 2) {
            // S = Q K^T: two 32-key blocks, four 16-deep steps each
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[j & 1].h, (MODE == 2 && j == 0) ? negm : s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(j + 1) & 1].h, q[j & 1].h, (MODE == 2 && j == 0) ? negm : s1, 0, 0, 0);
            }
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) { m = max3_raw(m, s0[e], s0[e + 1]); m = max3_raw(m, s1[e], s1[e + 1]); }
#pragma unroll
            for (int e = 0; e < 16; ++e) { s0[e] = exp2_raw(s0[e]); s1[e] = exp2_raw(s1[e]); }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { p[j].u[e] = pack_raw(s0[j * 8 + 2 * e], s0[j * 8 + 2 * e + 1]); p[2 + j].u[e] = pack_raw(s1[j * 8 + 2 * e], s1[j * 8 + 2 * e + 1]); }
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 32; e += 2) m = max3_raw(m, x[e], x[e + 1]);
#pragma unroll
            for (int e = 0; e < 32; ++e) x[e] = exp2_raw(x[e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) p[e >> 2].u[e & 3] = pack_raw(x[2 * e], x[2 * e + 1]);
        }
        if constexpr (MODE == 0 || MODE == 2) {
            // O += P V: two 32-wide halves of d = 64, four 16-key steps each
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, p[j].h, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(j + 1) & 1].h, p[j].h, o1, 0, 0, 0);
            }
        }
        if constexpr (MODE == 4) {
            // two half-steps with the roles of (s0, s1) and (t0v, t1v) swapped, so that no register moves are needed
#define HALF(SA0, SA1, SB0, SB1)                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
                SB0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[j & 1].h, j == 0 ? negm : SB0, 0, 0, 0);     \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) SA0[4 * j + e] = exp2_raw(SA0[4 * j + e]);                \
                p[j >> 1].u[2 * (j & 1)] = pack_raw(SA0[4 * j], SA0[4 * j + 1]); p[j >> 1].u[2 * (j & 1) + 1] = pack_raw(SA0[4 * j + 2], SA0[4 * j + 3]); \
                m = max3_raw(m, SA0[4 * j], SA0[4 * j + 1]); m = max3_raw(m, SA0[4 * j + 2], SA0[4 * j + 3]);            \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
                SB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(j + 1) & 1].h, q[j & 1].h, j == 0 ? negm : SB1, 0, 0, 0); \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) SA1[4 * j + e] = exp2_raw(SA1[4 * j + e]);                \
                p[2 + (j >> 1)].u[2 * (j & 1)] = pack_raw(SA1[4 * j], SA1[4 * j + 1]); p[2 + (j >> 1)].u[2 * (j & 1) + 1] = pack_raw(SA1[4 * j + 2], SA1[4 * j + 3]); \
                m = max3_raw(m, SA1[4 * j], SA1[4 * j + 1]); m = max3_raw(m, SA1[4 * j + 2], SA1[4 * j + 3]);            \
                __builtin_amdgcn_sched_barrier(0);                                                                       \
            }                                                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, p[j].h, o0, 0, 0, 0);                           \
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[(j + 1) & 1].h, p[j].h, o1, 0, 0, 0);                     \
            }
            HALF(s0, s1, t0v, t1v)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(negm), "+v"(q[0].h), "+v"(q[1].h));   // (or the second half's QK^T is the first one's, by CSE)
            HALF(t0v, t1v, s0, s1)
#undef HALF
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[(j >> 1) & 1].h, o1, 0, 0, 0);
                else       o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[(j >> 1) & 1].h, o0, 0, 0, 0);
                x[2 * j] = exp2_raw(x[2 * j]); x[2 * j + 1] = exp2_raw(x[2 * j + 1]);
                p[j >> 2].u[j & 3] = pack_raw(x[2 * j], x[2 * j + 1]);
                m = max3_raw(m, x[2 * j], x[2 * j + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = m;
    for (int e = 0; e < 16; ++e) s += s0[e] + s1[e] + o0[e] + o1[e] + t0v[e] + t1v[e];
    for (int e = 0; e < NX; ++e) s += x[e];
    for (int j = 0; j < 4; ++j) s += (float)p[j].h[0];
    sink[threadIdx.x] = s;
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, 16 * 8); (void)hipMalloc(&sink, 1024 * 4);
    const int iters = 4000;
    // s_memtime counts at the constant 100 MHz reference on gfx950; calibrate it against a known-length MFMA chain:
    // the 'mfma' body is 16 x 32 = 512 matrix cycles per step and one wave per SIMD cannot go faster than that.
    const char* names[5] = {"mfma  (16 MFMA)", "valu  (32 exp + 16 pack + 16 max3)", "chain (QK -> softmax -> PV, dependent)", "weave (same mix, independent)",
                            "pipe  (QK of j+1 woven with softmax of j)"};
    double base = 0;
    printf("%-42s %s\n", "body", "W=1        W=2        W=3        W=4      (ticks per step of ALL waves of a SIMD | matrix-pipe utilisation)");
    auto run = [&](auto kern, int mode) {
        printf("%-42s", names[mode]);
        for (int W = 1; W <= (mode == 4 ? 2 : 4); ++W) {
            const int steps = mode == 4 ? 2 : 1;                         // the 'pipe' body holds two steps per loop iteration
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(256 * W), 0, 0, d, sink, iters);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(16);
            (void)hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int w = 0; w < 4 * W; ++w) mx = h[w] > mx ? h[w] : mx;
            const double per = (double)mx / iters / steps;                       // ticks until every wave of the SIMD has done one step
            if (mode == 0 && W == 1) base = per;                         // = 512 matrix cycles
            const double util = mode == 1 ? 0.0 : W * base / per;
            printf(" %8.3f|%4.2f", per, util);
        }
        printf("\n");
    };
    run(k<0>, 0); run(k<1>, 1); run(k<2>, 2); run(k<3>, 3); run(k<4>, 4);
    printf("# 'mfma' at W=1 is 512 matrix cycles per step: ticks x %.1f = shader cycles\n", 512.0 / base);
    return 0;
}
