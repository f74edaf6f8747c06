// What does a kernel boundary cost on MI355X for the panel GEMM's launch shape?  Back-to-back launches of kernels that
// do (almost) nothing, by block count / block size / dynamic LDS / register budget, and of a kernel that only stores
// (the end-of-kernel write-back of dirty L2 lines).   hipcc --offload-arch=gfx950 -O3 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

extern __shared__ char smem[];
template <int T, int W>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(W, W))) void empty_kernel(int* out, int lds_touch) {
    if (lds_touch) smem[threadIdx.x] = 1;
    if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1;
}
// every block stores `bytes_per_block` (16 B per lane, coalesced) and exits
typedef float __attribute__((ext_vector_type(4))) f4;
template <int T>
__global__ __launch_bounds__(T) void store_kernel(f4* out, size_t vec_per_block, int sc) {
    f4* p = out + (size_t)blockIdx.x * vec_per_block;
    const f4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    for (size_t i = threadIdx.x; i < vec_per_block; i += T) {
        if (sc) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}

template <typename F> float time_launches(F f, int n, hipStream_t s) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) f();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* d; CK(hipMalloc(&d, 4));
    const int N = 2000;
    CK(hipFuncSetAttribute((const void*)empty_kernel<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)empty_kernel<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)empty_kernel<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int blocks : {1, 248, 1024, 4096})
        for (int lds : {0, 64 * 1024, 143 * 1024}) {
            float a = time_launches([&] { hipLaunchKernelGGL((empty_kernel<256, 8>), dim3(blocks), dim3(256), lds, s, d, 0); }, N, s);
            float b = time_launches([&] { hipLaunchKernelGGL((empty_kernel<256, 1>), dim3(blocks), dim3(256), lds, s, d, 0); }, N, s);
            float c = time_launches([&] { hipLaunchKernelGGL((empty_kernel<512, 2>), dim3(blocks), dim3(512), lds, s, d, 0); }, N, s);
            printf("empty  blocks %5d lds %6d B : 256 thr <=64 regs %6.2f us | 256 thr 512 regs %6.2f us | 512 thr 256 regs %6.2f us\n", blocks, lds, a, b, c);
        }
    // stores: 248 blocks x (77 MB / 248), plain and nontemporal; and the same bytes from 2048 blocks
    const size_t total = 77u * 1000 * 1000;
    f4* out; CK(hipMalloc(&out, total + 4096 * 16 * 4));
    for (int blocks : {248, 2048})
        for (int sc : {0, 1}) {
            const size_t vpb = total / 16 / blocks;
            float t = time_launches([&] { hipLaunchKernelGGL((store_kernel<256>), dim3(blocks), dim3(256), 0, s, out, vpb, sc); }, 300, s);
            printf("store  %4d blocks x %6.1f KB (%s): %6.2f us/launch = %5.2f TB/s\n", blocks, vpb * 16 / 1e3, sc ? "nontemporal" : "plain", t,
                   total / t / 1e6);
        }
    // a store kernel followed by an empty kernel vs alone: what the boundary adds behind dirty lines
    {
        const size_t vpb = total / 16 / 248;
        float t2 = time_launches([&] {
            hipLaunchKernelGGL((store_kernel<256>), dim3(248), dim3(256), 0, s, out, vpb, 0);
            hipLaunchKernelGGL((empty_kernel<256, 8>), dim3(248), dim3(256), 0, s, d, 0);
        }, 300, s);
        printf("store(248, plain) + empty(248): %6.2f us per pair\n", t2);
    }
    return 0;
}
