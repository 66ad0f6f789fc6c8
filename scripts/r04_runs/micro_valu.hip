// Issue-rate micro-benchmark of the VALU instructions the attention softmax and the GEGLU epilogue lean on (gfx950).
// One workgroup, W waves; every wave runs `iters` x 8 independent copies of one instruction between two s_memtime reads.
// Prints shader cycles per wave-instruction for 1 wave, 4 waves (one per SIMD) and 8 waves (two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o micro_valu.bin micro_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS(X) X(0, "v_exp_f32 %0, %0") X(1, "v_exp_f16 %0, %0") X(2, "v_fma_f32 %0, %0, %0, %0") X(3, "v_pk_fma_f32 %0, %0, %0, %0") \
               X(4, "v_pk_fma_f16 %0, %0, %0, %0") X(5, "v_rcp_f32 %0, %0") X(6, "v_cvt_pkrtz_f16_f32 %0, %0, %0") X(7, "v_max_f32 %0, %0, %0") \
               X(8, "v_pk_mul_f32 %0, %0, %0") X(9, "v_log_f32 %0, %0") X(10, "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1") X(11, "v_mov_b32 %0, %0") \
               X(12, "v_max3_f32 %0, %0, %0, %0") X(13, "v_cvt_pk_f16_f32 %0, %0, %0") X(14, "v_pk_max_f16 %0, %0, %0") X(15, "v_pk_mul_f16 %0, %0, %0")

template <int OP>
__global__ void k(unsigned long long* cyc, float* sink, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = (f2){0.001f * (threadIdx.x + j), 0.002f * j}; b[j] = a[j]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define X(N, S) if constexpr (OP == N) { if constexpr (N == 10) asm volatile(S : "+v"(a[j].x), "+v"(b[j].x)); \
                                         else if constexpr (N == 3 || N == 8) asm volatile(S : "+v"(a[j])); else asm volatile(S : "+v"(a[j].x)); }
            OPS(X)
#undef X
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += a[j].x + a[j].y + b[j].x;
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 64 * 8); hipMalloc(&sink, 4096 * 4);
    const int iters = 2000;
    const char* names[16];
#define X(N, S) names[N] = S;
    OPS(X)
#undef X
    printf("%-46s %10s %10s %10s   (shader cycles per wave-instruction; dual-issue rows: per PAIR)\n", "instruction", "1 wave", "4 waves", "8 waves");
    auto run = [&](auto kern, int op) {
        double r[3];
        int ws[3] = {1, 4, 8};
        for (int q = 0; q < 3; ++q) {
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * ws[q]), 0, 0, d, sink, iters);
            hipLaunchKernelGGL(kern, dim3(1), dim3(64 * ws[q]), 0, 0, d, sink, iters);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(8);
            hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost);
            unsigned long long mx = 0;
            for (int w = 0; w < ws[q]; ++w) mx = h[w] > mx ? h[w] : mx;
            r[q] = (double)mx / (iters * 8.0);
        }
        printf("%-46s %10.2f %10.2f %10.2f\n", names[op], r[0], r[1], r[2]);
    };
#define X(N, S) run(k<N>, N);
    OPS(X)
#undef X
    return 0;
}
