// Panel GEMM for the vision tower's big products, C = epilogue(A * B^T), bf16.
//
// Why a second GEMM kernel: with M = 32 * 197 = 6304 rows the 128x128 kernel (gemm.hip) reloads every operand
// panel once per 128 output columns, and its two-buffer pipeline exposes the L2 latency of every K step; both the
// L1/LDS-DMA path (~70 GB/s per CU) and the latency put it at ~30 % of the MFMA peak.  This kernel
//   * uses ONE block per CU with a tile of (16*MF) x (64*NF) chosen so that the whole product is a single round
//     of <= 256 blocks (e.g. 208 x 384 -> 31 x 8 = 248 blocks for N = 3072): 2x fewer operand bytes per flop;
//   * streams the activation panel A through a 4-stage LDS ring (global_load_lds, 128-byte rows, XOR swizzle)
//     with counted s_waitcnt vmcnt(N) and ONE raw s_barrier per 64-wide K step: three stages of loads in flight;
//   * reads the FROZEN weight operand straight from HBM/L2 into VGPRs in MFMA-fragment order (ffm_pack_b writes
//     that layout once at load time: every fragment is one contiguous, fully coalesced 1 KiB wave load), with a
//     4-deep register ring, so the weights never pass through LDS;
//   * runs 4 waves per block, ONE per SIMD, side by side along N (each wave owns 16*NF columns of all 16*MF rows,
//     up to 13 x 6 accumulator fragments = 312 registers, which only fit at one wave per SIMD: 256 AGPRs + VGPRs):
//     no two waves load the same weight fragment and the A fragments are the only LDS reads.
//
// VMEM ordering contract (the counted waits depend on it): every step issues, in this order,
//   [B fragments of half-step 2kt+3] [A ring stage kt+3: nA LDS-DMA ops] ... [B fragments of half-step 2kt+4]
// and asm memory fences + sched_barrier keep the compiler from moving them across step boundaries.
#include "gemm_panel.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int PW = 4;                      // waves per block: one per SIMD
constexpr int PT = PW * 64;
constexpr int PSTAGES = 4;                 // A ring depth (K64 stages)
typedef bf16x8 frag_t;

template <int MF, bool RK> struct PanelGeom {
    static constexpr int NB8 = 2 * MF + (RK ? 2 : 0);        // 8-row x 128-B DMA pieces per stage
    static constexpr int NI = NB8 / PW;                      // pieces per wave, the same for all waves (counted waits)
    static_assert(NB8 % PW == 0, "pick MF so that every wave issues the same number of DMA pieces");
    static constexpr int STAGE = NB8 * 1024;
    static constexpr int RING = PSTAGES * STAGE;
};

template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// a wave-uniform pointer the compiler can keep in SGPRs (inline-asm "s" operands must be scalar registers)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

__host__ __device__ constexpr int panel_stage_pitch(int nf) { return 16 * nf + 4; }          // floats
__host__ __device__ constexpr int panel_persist_bytes(int nf) { return PW * 16 * nf * 4; }  // bias [BN]

template <int MF, int NF, bool RK, int FL>
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_panel_kernel(ffm_gemm_args p) {
    using G = PanelGeom<MF, RK>;
    constexpr int BMp = 16 * MF, BNp = PW * 16 * NF, WN = 16 * NF;
    constexpr int flags = FL;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.N / BNp;
    const int tiles_m = (p.M + BMp - 1) / BMp;
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = logical / tiles_n, tn = logical % tiles_n;
    const int m0 = tm * BMp, n0 = tn * BNp, n0w = n0 + wave * WN;
    const int KT = p.K >> 6;

    // ---- A ring: per-lane source addresses of this wave's DMA pieces (piece = 8 rows x 128 B)
    const int rsub = lane >> 3, slot = lane & 7;
    const char* asrc[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        const int piece = wave + PW * i;
        const int chunk = (slot ^ rsub) << 4;
        if (RK && piece >= 2 * MF) {
            const int row = (piece - 2 * MF) * 8 + rsub;
            asrc[i] = reinterpret_cast<const char*>(p.rk) + (size_t)row * (size_t)p.K * 2 + chunk;
        } else {
            int grow = m0 + piece * 8 + rsub;
            grow = grow < p.M ? grow : p.M - 1;              // clamped rows are never stored
            asrc[i] = reinterpret_cast<const char*>(p.a) + (size_t)grow * (size_t)p.lda * 2 + chunk;
        }
    }
    // one DMA piece (i) of ring stage kt
    auto dma_piece = [&](int kt, int i) {
        if (p.G & 2) return;
        char* dst = smem + ((kt & 3) * G::STAGE) + wave * 1024 + i * (PW * 1024);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (size_t)kt * 128),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma = [&](int kt) {
#pragma unroll
        for (int i = 0; i < G::NI; ++i) dma_piece(kt, i);
    };

    // ---- B: fragment-packed weights, [N/16][K/32][64 lanes][8]; fragment (nf, hs) of this wave sits at
    // bbase + hs*1024 + boff[nf].  The loads are inline asm on purpose: the compiler's own waitcnt insertion
    // answers a mix of LDS-DMA and ordinary loads on vmcnt with s_waitcnt vmcnt(0), which would drain the
    // whole pipeline twice per step.  Every wait on vmcnt in the main loop is therefore counted by hand.
    const int K32 = p.K >> 5;
    const char* bbase = uniform_ptr(reinterpret_cast<const char*>(p.b_packed) + ((size_t)(n0w >> 4) * (size_t)K32) * 1024);
    int boff[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) boff[nf] = lane * 16 + nf * K32 * 1024;
    frag_t bq[4][NF];
    const int dbg = p.G;      // EXPERIMENT: bit0 = skip B loads, bit1 = skip A DMA, bit2 = skip MFMA
    auto loadB1 = [&](int hs, int nf, frag_t& dst) {
        if (dbg & 1) return;
        const char* sb = uniform_ptr(bbase + (size_t)hs * 1024);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(boff[nf]), "s"(sb) : "memory");
    };
    auto loadB = [&](int hs, frag_t (&dst)[NF]) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) loadB1(hs, nf, dst[nf]);
    };
    // after a counted wait: tie the fragments to this point so that no MFMA reading them is scheduled above it
    auto tieB = [&](frag_t (&b)[NF]) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) asm volatile("" : "+v"(b[nf]));
    };

    // Accumulators: 16*MF x 16*NF per wave = MF*NF fragments of 4 registers.  The first 64 live in AGPRs (all 256
    // of them), the rest in VGPRs; the MFMAs are inline asm with the accumulator tied in place (left to itself the
    // register allocator rotates MFMA destinations through extra registers, which a 312-register tile cannot afford).
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](auto IDX_, f32x4& c, const frag_t& a, const frag_t& b) {
        if constexpr (decltype(IDX_)::value < 64)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    };

    const int offA0 = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
    const int offA1 = offA0 ^ 64;
    // One half-step: MF A fragments (double-buffered by hand) x NF MFMAs each.  After fragment row mf the wave issues
    // ONE vector-memory instruction (issue(mf)): a VMEM issue stalls the wave until the texture path accepts it, and
    // with a single wave per SIMD a burst of 13 of them at the top of the step would stall the MFMA pipe with it.
    auto half = [&](const char* st, int off, const frag_t (&b)[NF], auto&& issue) {
        frag_t a[2];
        a[0] = *reinterpret_cast<const frag_t*>(st + off);
        static_for<MF>([&](auto MF_) {
            constexpr int mf = decltype(MF_)::value;
            if constexpr (mf + 1 < MF) a[(mf + 1) & 1] = *reinterpret_cast<const frag_t*>(st + (mf + 1) * 2048 + off);
            if (!(dbg & 4)) {
                static_for<NF>([&](auto NF_) {
                    constexpr int nf = decltype(NF_)::value;
                    mma(std::integral_constant<int, mf * NF + nf>{}, acc[mf][nf], a[mf & 1], b[nf]);
                });
            }
            issue(MF_);
        });
    };
    static_assert(NF + G::NI <= MF, "one VMEM slot per fragment row");

    // ---- prologue: ring stages 0..2 and the B fragments of half-steps 0..2, drained once before the loop
    dma(0);
    dma(1);
    dma(2);
    fence();
    loadB(0, bq[0]);
    loadB(1, bq[1]);
    loadB(2, bq[2]);
    fence();

    float* Bias = reinterpret_cast<float*>(smem + G::RING);
    for (int i = tid; i < BNp; i += PT) Bias[i] = (flags & FFM_EPI_BIAS) ? p.bias[n0 + i] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // VMEM issue order of step kt (nA = G::NI pieces, the same on every wave):
    //   first half : B1 = the NF fragments of half-step 2kt+3, then the nA pieces of A stage kt+3
    //   second half: B2 = the NF fragments of half-step 2kt+4
    // Younger ops behind each counted wait, by steps left rem = KT - kt (B1 exists while rem >= 2, the A stage while
    // rem >= 4, B2 while rem >= 3):
    //                                                                             rem >= 4  | rem 3     | rem 2 | rem 1
    //   first half  needs B2(kt-2):  ops(kt-1)                                  = 2NF + nA  | 2NF + nA  | 2NF   | NF
    //   second half needs B1(kt-1):  A(kt-1) + B2(kt-1) + B1(kt) + A(kt)        = 2NF + 2nA | 2NF + nA  | 2NF   | 0
    //   end of step needs A stage kt+1 (issued in step kt-2): B2(kt-2)+ops(kt-1)+ops(kt) = 5NF + 2nA | 5NF + nA | 4NF | -
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr int nA = G::NI;
    auto step = [&](int kt, auto P_, auto TAIL_) {
        constexpr int P = decltype(P_)::value;
        constexpr bool TAIL = decltype(TAIL_)::value != 0;      // tail: the last four steps, guarded by rem
        const int rem = KT - kt;
        const char* st = smem + (kt & 3) * G::STAGE;
        if (!TAIL || rem >= 3) wait_vm<2 * NF + nA>();
        else if (rem == 2) wait_vm<2 * NF>();
        else wait_vm<NF>();
        tieB(bq[2 * P]);
        half(st, offA0, bq[2 * P], [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < NF) {
                if (!TAIL || rem >= 2) loadB1(2 * kt + 3, j, bq[(2 * P + 3) & 3][j]);
            } else if constexpr (j < NF + nA) {
                if (!TAIL || rem >= 4) dma_piece(kt + 3, j - NF);
            }
        });
        fence();
        if (!TAIL || rem >= 4) wait_vm<2 * NF + 2 * nA>();
        else if (rem == 3) wait_vm<2 * NF + nA>();
        else if (rem == 2) wait_vm<2 * NF>();
        else wait_vm<0>();
        tieB(bq[2 * P + 1]);
        half(st, offA1, bq[2 * P + 1], [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < NF) {
                if (!TAIL || rem >= 3) loadB1(2 * kt + 4, j, bq[2 * P][j]);
            }
        });
        fence();
        if (!TAIL || rem >= 2) {
            if (!TAIL || rem >= 4) wait_vm<5 * NF + 2 * nA>();
            else if (rem == 3) wait_vm<5 * NF + nA>();
            else wait_vm<4 * NF>();
            block_barrier();
        }
    };
    int kt = 0;
    for (; kt < KT - 4; kt += 2) {                               // steady state: no guards, no branches
        step(kt, I0{}, I0{});
        step(kt + 1, I1{}, I0{});
    }
    for (; kt < KT; kt += 2) {                                   // last four steps
        step(kt, I0{}, I1{});
        step(kt + 1, I1{}, I1{});
    }
    // the asm MFMAs are invisible to the hazard recogniser: let the last ones retire before the accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();                                                  // ring is free: epilogue staging

    // ---------------- epilogue: per wave, 32-row groups through a private LDS stage ----------------
    constexpr int PITCH = panel_stage_pitch(NF);
    constexpr int CPR = 2 * NF;                       // 8-column chunks per row of the wave's slab
    float* Cw = reinterpret_cast<float*>(smem) + wave * (32 * PITCH);
    const int frow = lane & 15, fgrp = lane >> 4;
    bf16_t* C = reinterpret_cast<bf16_t*>(p.c);
    constexpr int NRG = (MF + 1) / 2;
    constexpr int PF = 3;                             // row groups of residual / aux rows kept in flight
    bf16x8 rres[PF][NF];
    auto load_res = [&](int rg, bf16x8 (&dst)[NF]) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR;
            const int gm = m0 + rg * 32 + row;
            if (gm < p.M && rg * 32 + row < BMp)
                dst[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldc + n0w + ch * 8);
        }
    };
    if constexpr ((flags & FFM_EPI_RESIDUAL) != 0) {
#pragma unroll
        for (int g = 0; g < PF && g < NRG; ++g) load_res(g, rres[g]);
    }
    // accumulator fragment -> stage: explicit ds_write_b32 (data straight from the AGPR / VGPR the MFMAs left it in)
    const uint32_t cw_lane = (uint32_t)(uintptr_t)(Cw + fgrp * 4 * PITCH + frow);
    static_for<NRG>([&](auto RG_) {
        constexpr int rg = decltype(RG_)::value;
        static_for<2 * NF * 4>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int q = t / (NF * 4), nf = (t / 4) % NF, e = t % 4, mf = 2 * rg + q;
            constexpr int off = ((q * 16 + e) * PITCH + nf * 16) * 4;
            const uint32_t cwl = cw_lane;               // (named outside the if constexpr so that the lambda captures them)
            const f32x4(&accr)[MF][NF] = acc;
            if constexpr (mf < MF) {
                if constexpr (mf * NF + nf < 64)
                    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(cwl), "a"(accr[mf][nf][e]), "n"(off) : "memory");
                else
                    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(cwl), "v"(accr[mf][nf][e]), "n"(off) : "memory");
            }
        });
        fence();
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR;
            const int gm = m0 + rg * 32 + row;
            const bool ok = gm < p.M && rg * 32 + row < BMp;
            float v[8];
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&Cw[row * PITCH + ch * 8]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&Cw[row * PITCH + ch * 8 + 4]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bias[wave * WN + ch * 8]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bias[wave * WN + ch * 8 + 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] = c0[c] + b0[c]; v[4 + c] = c1[c] + b1[c]; }
            if constexpr ((flags & FFM_EPI_RESIDUAL) != 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += (float)rres[rg % PF][i][c];
            }
            if (ok) Vec8<bf16_t>::store(C + (size_t)gm * p.ldc + n0w + ch * 8, v);
        }
        if constexpr ((flags & FFM_EPI_RESIDUAL) != 0) {
            if constexpr (rg + PF < NRG) load_res(rg + PF, rres[rg % PF]);
        }
        fence();
    });
}

template <int MF, int NF, bool RK, int FL>
int launch_panel(const ffm_gemm_args& a, hipStream_t s) {
    using G = PanelGeom<MF, RK>;
    const int tiles = ((a.M + 16 * MF - 1) / (16 * MF)) * (a.N / (PW * 16 * NF));
    int lds = G::RING + panel_persist_bytes(NF);
    static bool done = false;                         // one per instantiation
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_kernel<MF, NF, RK, FL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    hipLaunchKernelGGL((gemm_panel_kernel<MF, NF, RK, FL>), dim3(tiles), dim3(PT), lds, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// dst[((n16 * K/32 + k32) * 64 + lane) * 8 + i] = src[(n16*16 + (lane & 15)) * ld + k32*32 + (lane >> 4)*8 + i]
__global__ __launch_bounds__(256) void pack_b_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                                     int ld) {
    const size_t total = (size_t)(N >> 4) * (K >> 5) * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t frag = i >> 6;
        const int k32 = (int)(frag % (size_t)(K >> 5)), n16 = (int)(frag / (size_t)(K >> 5));
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)(n16 * 16 + (lane & 15)) * ld + k32 * 32 + (lane >> 4) * 8);
        *reinterpret_cast<bf16x8*>(dst + i * 8) = v;
    }
}

}  // namespace

// Cost model (both kernels are bound by the bytes a CU pulls through its L1 / LDS-DMA path): rounds x operand rows
// loaded per K step.  The 128x128 kernel runs two blocks per CU, so a CU with two tiles loads 2 x 256 rows.
int ffm_panel_select(int M, int N, int K, int flags, int rank, int dtype, bool packed) {
    if (!packed || dtype != FFM_BF16 || K % 128 != 0 || K < 512) return -1;
    if (flags & ~(FFM_EPI_BIAS | FFM_EPI_RESIDUAL)) return -1;
    (void)rank;
    if (const char* f = getenv("FFM_PANEL_FORCE")) {          // debugging aid: force one configuration
        const int c = atoi(f);
        if (c >= 0 && c < FFM_PANEL_NCFG && N % (64 * FFM_PANEL_CFGS[c].nf) == 0 &&
            FFM_PANEL_CFGS[c].rankop == ((flags & FFM_EPI_RANKOP) != 0))
            return c;
    }
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    long best = ((t128 + 255) / 256) * 256;
    int pick = -1;
    for (int c = 0; c < FFM_PANEL_NCFG; ++c) {
        const int bm = 16 * FFM_PANEL_CFGS[c].mf, bn = 64 * FFM_PANEL_CFGS[c].nf;
        if (N % bn || FFM_PANEL_CFGS[c].rankop != ((flags & FFM_EPI_RANKOP) != 0)) continue;
        const long blocks = (long)((M + bm - 1) / bm) * (N / bn);
        const long cost = ((blocks + 255) / 256) * (bm + bn);
        if (cost < best) { best = cost; pick = c; }
    }
    return pick;
}

#define PANEL_CASE(F, RKB)                                                           \
    case F:                                                                          \
        switch (cfg) {                                                               \
            case 1: return launch_panel<16, 4, RKB, F>(a, s);                        \
            case 2: return launch_panel<10, 2, RKB, F>(a, s);                        \
        }                                                                            \
        return FFM_EINVAL;

int ffm_panel_launch(const ffm_gemm_args& a_, int cfg, hipStream_t s) {
    ffm_gemm_args a = a_;
    if (const char* f = getenv("FFM_PANEL_DBG")) a.G = atoi(f); else a.G = 0;
    if (((uintptr_t)a.b_packed & 15) || a.ldc % 8) return FFM_EINVAL;
    switch (a.flags) {
        PANEL_CASE(0, false)
        PANEL_CASE(FFM_EPI_BIAS, false)
        PANEL_CASE(FFM_EPI_BIAS | FFM_EPI_RESIDUAL, false)
    }
    return FFM_EINVAL;
}

extern "C" int ffm_pack_b(const void* src, void* dst, int N, int K, int ld, void* stream) {
    if (!src || !dst || N <= 0 || K <= 0 || N % 16 || K % 32 || ld < K || ld % 8) return FFM_EINVAL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return FFM_EINVAL;
    const size_t total = (size_t)(N >> 4) * (K >> 5) * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_b_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, N, K,
                       ld);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
