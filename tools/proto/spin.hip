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

// the same with a register footprint: a clobbered high VGPR makes the kernel allocate that many registers per wave (a text-tower
// GEMM's are 150-240), so that a block of it cannot share a SIMD's register file with anything large
#define SPIN_FAT(NAME, REG)                                                                      \
    __global__ void NAME(unsigned ticks, unsigned* sink) {                                        \
        extern __shared__ unsigned sm[];                                                           \
        asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                  \
        unsigned n = 0;                                                                            \
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {                                    \
            __builtin_amdgcn_s_sleep(8);                                                           \
            ++n;                                                                                   \
        }                                                                                          \
        if (n == 0xffffffffu) sink[0] = sm[threadIdx.x & 15];                                      \
    }
SPIN_FAT(spin_v32, "v31")
SPIN_FAT(spin_v64, "v63")
SPIN_FAT(spin_v96, "v95")
SPIN_FAT(spin_v128, "v127")
SPIN_FAT(spin_v168, "v167")
SPIN_FAT(spin_v240, "v239")

// vgprs: 32 | 64 | 96 | 128 | 168 | 240
extern "C" int spin_fat_launch(int blocks, int threads, int lds, int usec, void* sink, void* stream, int vgprs) {
    void (*k)(unsigned, unsigned*) = vgprs <= 32 ? spin_v32 : vgprs <= 64 ? spin_v64 : vgprs <= 96 ? spin_v96 : vgprs <= 128 ? spin_v128
                                     : vgprs <= 168 ? spin_v168 : spin_v240;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (unsigned)(usec * 100), (unsigned*)sink);
    return (int)hipGetLastError();
}

extern "C" int spin_launch(int blocks, int threads, int lds, int usec, void* sink, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (unsigned)(usec * 100), (unsigned*)sink);
    return (int)hipGetLastError();
}
