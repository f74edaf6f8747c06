// Prototype (not part of the library): main loop of an 8-wave panel GEMM for the N = 768 products.
// Block = 512 threads = 2 (M) x 4 (N) waves, tile (32 MFW) x 128, every operand through a 3-stage LDS ring by DMA
// (A rows swizzled as in gemm_panel_impl.h, B as packed MFMA fragments), one barrier per K64 step.
// Question it answers: does a second wave per SIMD overlap the texture-path / LDS work with the MFMAs?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MFW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void p8_kernel(const bf16_t* __restrict__ A,
                                                                                           const bf16_t* __restrict__ Bp,
                                                                                           bf16_t* __restrict__ C, int M, int N, int K,
                                                                                           unsigned long long* stamps) {
    constexpr int BM = 32 * MFW, NPA = BM / 8, NPB = 16, NPC = NPA + NPB, STAGE = NPC * 1024, NS = 3;
    constexpr int NI = (NPC + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = N / 128;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * 128;
    const int KT = K >> 6, K32 = K >> 5;
    const int frow = lane & 15, fgrp = lane >> 4;
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();

    // this wave's DMA pieces q = wave, wave + 8, ...: source pointer and per-K64 byte step
    const char* src[NI];
    int kstep[NI];
    int mine = 0;
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = wave + 8 * i;
        if (q < NPA) {
            int row = m0 + q * 8 + rsub;
            row = row < M ? row : M - 1;
            src[i] = reinterpret_cast<const char*>(A) + (size_t)row * K * 2 + ((slot ^ rsub) << 4);
            kstep[i] = 128;
            ++mine;
        } else if (q < NPC) {
            const int f = q - NPA, n16 = f >> 1, h = f & 1;            // fragment (n16, half) of the stage
            src[i] = reinterpret_cast<const char*>(Bp) + ((size_t)((n0 >> 4) + n16) * K32 + h) * 1024 + lane * 16;
            kstep[i] = 2048;
            ++mine;
        } else {
            src[i] = nullptr;
            kstep[i] = 0;
        }
    }
    auto dma = [&](int kt) {
        char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = wave + 8 * i;
            if (q < NPC)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * kstep[i]),
                                                 (__attribute__((address_space(3))) void*)(st + q * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[MFW][2];
#pragma unroll
    for (int i = 0; i < MFW; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    dma(0);
    dma(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();

    const int offA0 = (wm * MFW * 16 + frow) * 128 + ((fgrp ^ (lane & 7)) << 4);
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 2 < KT) dma(kt + 2);
        const char* st = smem + (kt % NS) * STAGE;
        const uint32_t sa = (uint32_t)(uintptr_t)st;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 a[MFW], b[2];
            const uint32_t ao = sa + (uint32_t)(h ? (offA0 ^ 64) : offA0);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf)
                asm volatile("ds_read_b128 %0, %1" : "=v"(b[nf]) : "v"(sa + NPA * 1024 + ((wn * 2 + nf) * 2 + h) * 1024 + lane * 16) : "memory");
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf)
                asm volatile("ds_read_b128 %0, %1" : "=v"(a[mf]) : "v"(ao + mf * 2048) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf) asm volatile("" : "+v"(a[mf]));
            asm volatile("" : "+v"(b[0]), "+v"(b[1]));
#pragma unroll
            for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mf], b[nf], acc[mf][nf], 0, 0, 0);
        }
        // stage kt + 1 (issued one step ago) must have landed; the pieces of stage kt + 2 may stay in flight
        if (kt + 2 < KT) {
            if (mine == NI) wait_vm<NI>(); else wait_vm<NI - 1>();
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
    }
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    // plain store of the tile (timing prototype: no epilogue)
#pragma unroll
    for (int mf = 0; mf < MFW; ++mf)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + (wm * MFW + mf) * 16 + fgrp * 4 + e, col = n0 + wn * 32 + nf * 16 + frow;
                if (row < M) C[(size_t)row * N + col] = (bf16_t)acc[mf][nf][e];
            }
    if (stamps && tid == 0) stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
}

extern "C" int p8_run(const void* A, const void* Bp, void* C, int M, int N, int K, int mfw, void* stamps, void* stream) {
    const int tiles = ((M + 32 * mfw - 1) / (32 * mfw)) * (N / 128);
#define RUN(MFW)                                                                                                        \
    {                                                                                                                   \
        constexpr int lds = 3 * ((32 * MFW) / 8 + 16) * 1024;                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(p8_kernel<MFW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL((p8_kernel<MFW>), dim3(tiles), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)A,        \
                           (const bf16_t*)Bp, (bf16_t*)C, M, N, K, (unsigned long long*)stamps);                        \
    }
    if (mfw == 5) RUN(5) else if (mfw == 6) RUN(6) else return -1;
    return (int)hipGetLastError();
}
