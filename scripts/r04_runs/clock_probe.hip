// A one-wave kernel that stores (s_memtime, s_memrealtime) - the free-running shader-clock counter and the constant 100 MHz
// counter - so that two probes around a stretch of work on the same stream give the AVERAGE shader clock over that stretch
// (scripts/r04_runs/forward_clock.py).   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclock_probe.so clock_probe.hip
#include <hip/hip_runtime.h>
__global__ void probe_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime(); out[1] = __builtin_amdgcn_s_memrealtime(); }
}
extern "C" void clock_probe(void* out, void* stream) { hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out); }
