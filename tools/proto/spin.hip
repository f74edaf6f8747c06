// Stand-alone probe (tools/side_proxy.py): a kernel that occupies `blocks` workgroups of `threads` threads (and `lds` bytes
// of LDS each) for `usec` microseconds on the constant 100 MHz clock - a stand-in for side-stream work of a given grid
// shape, to price what its PLACEMENT costs the vision chain without its memory traffic.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/proto/libspin.so tools/proto/spin.hip
#include <hip/hip_runtime.h>

__global__ void spin_kernel(unsigned ticks, unsigned* sink) {
    extern __shared__ unsigned sm[];
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (n == 0xffffffffu) sink[0] = sm[threadIdx.x & 15];
}

extern "C" int spin_launch(int blocks, int threads, int lds, int usec, void* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (unsigned)(usec * 100), (unsigned*)sink);
    return (int)hipGetLastError();
}
