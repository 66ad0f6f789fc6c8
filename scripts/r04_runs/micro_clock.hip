// What clock does the chip sustain under matrix-pipe load?  Every workgroup (256 threads = one wave per SIMD, or 512 = two) runs
// `iters` steps of a body from micro_overlap.hip - 'mfma' (16 MFMAs), 'weave' (16 MFMAs with the softmax VALU mix between them), 'valu'
// (the VALU mix alone) - and reads BOTH counters around the loop: s_memtime (shader clock) and s_memrealtime (constant 100 MHz).
// shader ticks / real-time ticks x 100 MHz = the clock the CU actually ran at.  Launched on 1 workgroup, on 256 (one per CU) and on
// 2048 (the whole chip, 8 per CU in turn), for ~10 ms each so that power management has time to settle.
//   hipcc --offload-arch=gfx950 -O3 -o micro_clock.bin micro_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float exp2_raw(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ unsigned pack_raw(float a, float b) { unsigned r; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
union Frag { h8 h; unsigned u[4]; };

template <int MODE>   // 0 mfma, 1 valu, 2 weave
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    Frag q[2], kf[2], p[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) { q[j & 1].h[e] = (_Float16)(0.01f * ((lane + e + j) & 7)); kf[j & 1].h[e] = (_Float16)(0.02f * ((lane * 3 + e + j) & 7)); p[j].h[e] = (_Float16)0.5f; }
    f16v o0, o1;
    for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }
    float x[32], m = -3.f;
    for (int e = 0; e < 32; ++e) x[e] = -0.01f * (lane + e);
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(q[0].h), "+v"(q[1].h));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (MODE != 1) {
                if (j & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[(j >> 1) & 1].h, o1, 0, 0, 0);
                else       o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[j & 1].h, q[(j >> 1) & 1].h, o0, 0, 0, 0);
            }
            if constexpr (MODE != 0) {
                x[2 * j] = exp2_raw(x[2 * j]); x[2 * j + 1] = exp2_raw(x[2 * j + 1]);
                p[j >> 2].u[j & 3] = pack_raw(x[2 * j], x[2 * j + 1]);
                m = max3_raw(m, x[2 * j], x[2 * j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = m;
    for (int e = 0; e < 16; ++e) s += o0[e] + o1[e];
    for (int e = 0; e < 32; ++e) s += x[e];
    for (int j = 0; j < 4; ++j) s += (float)p[j].h[0];
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 4095] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    unsigned long long* d; float* sink;
    (void)hipMalloc(&d, 4096 * 16); (void)hipMalloc(&sink, 4096 * 4);
    const char* names[3] = {"mfma", "valu", "weave"};
    printf("%-6s %6s %6s %10s %12s %12s %10s\n", "body", "waves", "WGs", "iters", "shader tk/step", "us/step", "clock GHz");
    auto run = [&](auto kern, int mode, int threads, int grid, int iters) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, d, sink, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * grid);
        (void)hipMemcpy(h.data(), d, 16 * grid, hipMemcpyDeviceToHost);
        std::vector<double> clk, tk;
        for (int b = 0; b < grid; ++b) { clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); tk.push_back((double)h[2 * b] / iters); }
        std::sort(clk.begin(), clk.end()); std::sort(tk.begin(), tk.end());
        const double med_clk = clk[grid / 2], med_tk = tk[grid / 2];
        printf("%-6s %6d %6d %10d %12.1f %12.4f %10.3f   (clock min %.3f max %.3f)\n", names[mode], threads / 256, grid, iters, med_tk, med_tk / (med_clk * 1e3), med_clk, clk.front(), clk.back());
    };
    for (int threads : {256, 512})
        for (int grid : {1, 256, 2048}) {
            const int iters = grid == 2048 ? 5000 : 40000;       // ~10 ms at 512 cycles per step
            run(k<0>, 0, threads, grid, iters); run(k<1>, 1, threads, grid, iters); run(k<2>, 2, threads, grid, iters);
        }
    return 0;
}
