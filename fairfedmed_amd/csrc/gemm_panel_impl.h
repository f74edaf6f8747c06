// Panel GEMM for the vision tower's big products, C = epilogue(A * B^T), bf16 (instantiated by gemm_panel*.hip).
//
// Why a second GEMM kernel: with M = 32 * 197 = 6304 rows the 128x128 kernel (gemm.hip) reloads every operand
// panel once per 128 output columns through the CU's 64 B/clk texture path, which is as busy as the MFMA pipe at
// that tile size, and its two-buffer pipeline exposes the L2 latency of every K step.  This kernel
//   * uses ONE block per CU with a tile of (16*MF) x (64*NF) chosen so that the whole product is a single round
//     of <= 256 blocks (208 x 384 -> 31 x 8 = 248 blocks for N = 3072): half the operand bytes per flop;
//   * streams the activation panel A through a 4-stage LDS ring (global_load_lds, 128-byte rows, XOR swizzle)
//     with counted s_waitcnt vmcnt(N) and ONE raw s_barrier per 64-wide K step: three stages of loads in flight;
//   * reads the FROZEN weight operand straight from HBM/L2 into VGPRs in MFMA-fragment order (ffm_pack_b writes
//     that layout once at load time: every fragment is one contiguous, fully coalesced 1 KiB wave load), with a
//     4-deep register ring, so the weights never pass through LDS;
//   * runs 4 waves per block, ONE per SIMD, side by side along N (each wave owns 16*NF columns of all 16*MF rows,
//     up to 13 x 6 accumulator fragments = 312 registers, which only fit at one wave per SIMD: 256 AGPRs + VGPRs):
//     no two waves load the same weight fragment and the A fragments are the only LDS reads;
//   * spreads its vector-memory instructions over the MFMA stream, one after each fragment row: a VMEM issue
//     stalls the wave until the texture path accepts it, and with one wave per SIMD a burst of them would stall
//     the matrix pipe too (measured: 0.7 us per K step before, 0.1 us after).
// Measured on MI355X (tools/bench_panel.py): the main loop alone runs at ~2.0 PFLOP/s (the MFMA rate at the clock
// the chip sustains); prologue, epilogue and the output burst of a single-round kernel are NOT overlapped with it.
//
// VMEM ordering contract (the counted waits depend on it): see the table in front of the main loop.
#pragma once
#include "gemm_panel.h"
#include <type_traits>

namespace ffm_panel {

// Waves per block (template parameter PWV of the kernel): 4 = one per SIMD, each owning 16*NF columns of all rows and up
// to 312 accumulator registers; 8 = two per SIMD (<= 256 registers each) side by side along N in the same way, so that
// one wave's MFMAs issue while its SIMD partner sits in an LDS / VMEM issue slot or in the epilogue's VALU work.
constexpr int PSTAGES = 4;                 // A ring depth (K64 stages)
typedef bf16x8 frag_t;

template <int MF, bool RK, int PWV> struct PanelGeom {
    static constexpr int NB8 = 2 * MF + (RK ? 2 : 0);        // 8-row x 128-B DMA pieces per stage (+16 rank rows)
    // pieces per wave, the same for all waves (the counted waits depend on it): when NB8 is not a multiple of the wave
    // count, the waves that run out repeat their previous piece (same source rows, same LDS destination: harmless)
    static constexpr int NI = (NB8 + PWV - 1) / PWV;
    static_assert(NI >= 2 || NB8 % PWV == 0, "a repeated piece needs a previous one");
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

// Diagnostic build only (-DFFM_PANEL_STAMPS, tools/panel_stamps.py): wave 0 of every block writes {shader clock,
// 100 MHz real-time clock} pairs at phase boundaries into the buffer passed in the (otherwise unused) `ts` field.
#ifdef FFM_PANEL_STAMPS
#define FFM_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (p.ts && tid == 0) {                                                                              \
            unsigned long long* sb__ = (unsigned long long*)p.ts + ((size_t)blockIdx.x * 12 + (i)) * 2;       \
            sb__[0] = __builtin_amdgcn_s_memtime();                                                          \
            sb__[1] = __builtin_amdgcn_s_memrealtime();                                                      \
        }                                                                                                    \
    } while (0)
#else
#define FFM_STAMP(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LGRAD: a V element as the 16-bit pair {hi, lo = v - hi} in one dword (split ONCE per block, not once per wave and row group)
__device__ __forceinline__ uint32_t lg_split(float v) {
    const bf16_t h = (bf16_t)v, l = (bf16_t)(v - (float)h);
    return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
}
// LGRAD: the 2 NF transposed reads of one [32][16 NF] image (rows 8 g + q / + 4 of the lane's group, 16 columns each; the
// two images are interleaved row by row: row stride 2 TROWX, the second image TROWX behind the first), all on
// ONE address register with immediate offsets (as computed addresses the compiler kept all of them live across the row
// groups and spilled them - with a vmcnt(0) in front of every reload)
typedef __attribute__((ext_vector_type(2))) uint32_t panel_u32x2;
template <int NFX, int TROWX, int TEN, int T = 0> __device__ __forceinline__ void lg_tr_reads(panel_u32x2 (&xr)[NFX][2], uint32_t a) {
    if constexpr (T < NFX * 2) {
        constexpr int cf = T / 2, h = T % 2;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xr[cf][h]) : "v"(a), "n"(TEN * TROWX + h * 4 * 2 * TROWX + cf * 32) : "memory");
        lg_tr_reads<NFX, TROWX, TEN, T + 1>(xr, a);
    }
}
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
// Diagnostic builds only (tools/panel_stamps.py): drop parts of the main loop to price them.  1: no weight-fragment
// loads, 2: no LDS-DMA of the activation ring, 4: no MFMAs, 8: no LDS fragment reads (results are garbage).
#ifndef FFM_PANEL_ABL
#define FFM_PANEL_ABL 0
#endif

__host__ __device__ constexpr int stage_pitch(int nf) { return 16 * nf + 4; }                // floats
// persistent LDS behind the ring: bias [BN] f32 | (RANKOP) LoRA tile [BN][32] bf16 | lora_S [256] + its column sums [16]
// f32 | row groups [BM]
// | (LNIN) c [BN] f32, row mean / rstd [BM] f32 each
// | (LGRAD) the forward ts rows of the other adapter [BM][r] f32 (room for r = 16)
// | (LNB_STAT) W gamma [BN], W beta + b [BN] f32      | (LNB_APPLY: the LNIN block - gamma in c's place - and c1, c2 [BM] f32)
__host__ __device__ constexpr int persist_bytes(int mf, int nf, bool rk, bool lnin = false, int pw = 4, bool lgrad = false, bool lnbs = false,
                                                bool lnba = false) {
    // (lnbs: no persistent operands since the sums against W gamma and d moved into the consumer's rank operand)
    return pw * 16 * nf * 4 + (rk ? pw * 16 * nf * 64 + (256 + 16) * 4 + 16 * mf * 4 : 0) +
           (lnin ? pw * 16 * nf * 4 + 2 * 16 * mf * 4 : 0) + (lgrad ? 16 * mf * 64 : 0) + (lnba ? 2 * 16 * mf * 4 : 0);
}

// Two blocks per CU (two waves per SIMD) for the 128 x 256 plain tile: 128 accumulator registers and a 64 KiB ring
// leave room for it, and the second wave issues MFMAs while the first sits in a VMEM / LDS issue slot.
template <int MF, int NF, bool RK, int PWV> constexpr int panel_waves_per_eu() {
    return (PWV == 8 || (MF == 8 && NF == 4 && !RK)) ? 2 : 1;
}

// KS (K split, needs PWV = 8): the eight waves are 4 column slabs x 2 K halves.  Waves 0-3 take the first 32 of every
// 64-deep K step of the ring stage, waves 4-7 the second 32, each with accumulators for the block's WHOLE slab (the
// 128-column tiles: 10 x 2 fragments = 80 registers), summed through LDS behind the loop; the epilogue runs on waves
// 0-3.  Unlike eight column slabs (section 4.4 of DESIGN.md: no gain on these tiles) this keeps the LDS fragment reads
// and the weight-fragment loads per step exactly those of the 4-wave kernel - every A half-fragment is read by four
// waves, every B fragment loaded once - while each SIMD gets a second wave whose MFMAs fill the other's issue stalls.
template <int MF, int NF, bool RK, int FL, int PWV = 4, int KS = 0>
__global__ __launch_bounds__(PWV * 64)
__attribute__((amdgpu_waves_per_eu(panel_waves_per_eu<MF, NF, RK, PWV>(), panel_waves_per_eu<MF, NF, RK, PWV>()))) void gemm_panel_kernel(ffm_gemm_args p) {
    constexpr int PW = PWV, PT = PWV * 64;
    static_assert(!KS || PWV == 8, "K split: 4 column slabs x 2 K halves");
    constexpr int CW = KS ? PW / 2 : PW;                     // column slabs (waves side by side along N)
    using G = PanelGeom<MF, RK, PWV>;
    constexpr int BMp = 16 * MF, BNp = CW * 16 * NF, WN = 16 * NF;
    constexpr int flags = FL;
    static_assert(!RK || (flags & FFM_EPI_LORA), "RANKOP rides on the LoRA epilogue");
    // LGRAD (dX of c_proj): the two rank-r gradient reductions whose [rows x N] operand this launch holds in registers -
    // dB of the other adapter from the rows it stores (c = dL/d pre) and dA of its own from quick_gelu(aux) - are formed
    // per row tile in the output epilogue (see there) instead of by two kernels that read 2 x M x N elements once more
    constexpr bool LGRAD = (FL & FFM_EPI_LGRAD) != 0;
    static_assert(!LGRAD || (RK && (FL & FFM_EPI_DGELU) && !KS), "LGRAD rides on the DGELU + RANKOP epilogue");
    // LayerNorm backward folded into the two dX products of the MLP (include/ffm_hip.h, FFM_EPI_LNB_*): LNBS - this launch
    // (dX of c_proj) leaves the two row sums per column tile; LNBA - this launch (dX of c_fc) applies
    // rstd (gamma g_h - c1/K - xhat c2/K) + res to its rows instead of storing g_h
    constexpr bool LNBS = (FL & FFM_EPI_LNB_STAT) != 0, LNBA = (FL & FFM_EPI_LNB_APPLY) != 0;
    static_assert(!LNBS || LGRAD, "LNB_STAT rides on the LGRAD epilogue (its chunk loop holds dpre and pre)");
    static_assert(!LNBA || (!KS && !(FL & (FFM_EPI_RESIDUAL | FFM_EPI_DGELU | FFM_EPI_GELU | FFM_EPI_LNIN | FFM_EPI_ROWSTATS | FFM_EPI_BIAS))),
                  "LNB_APPLY: a dX epilogue without anything else in it (FairLoRA: c_fc; plain: the in-projection)");
    static_assert(RK || !(flags & FFM_EPI_LORA), "the panel kernel only has the in-kernel (RANKOP) LoRA epilogue");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FFM_STAMP(0);
    const int tiles_n = p.N / BNp;
    const int tiles_m = (p.M + BMp - 1) / BMp;
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = logical / tiles_n, tn = logical % tiles_n;
    const int colw = KS ? (wave & (CW - 1)) : wave;           // column slab of this wave
    const int kg = KS ? wave / CW : 0;                        // K half (K split only)
    const int m0 = tm * BMp, n0 = tn * BNp, n0w = n0 + colw * WN;
    const int KT = p.K >> 6;
    const int frow = lane & 15, fgrp = lane >> 4;

    // ---- A ring: per-lane source addresses of this wave's DMA pieces (piece = 8 rows x 128 B)
    const int rsub = lane >> 3, slot = lane & 7;
    const char* asrc[G::NI];
    int apiece[G::NI];
#pragma unroll
    for (int i = 0; i < G::NI; ++i) {
        int piece = wave + PW * i;
        if (piece >= G::NB8) piece -= PW;                    // (wave-uniform) out of pieces: repeat the previous one
        apiece[i] = piece;
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
    auto dma_piece = [&](int kt, int i) {                    // piece i of ring stage kt
        char* dst = smem + ((kt & 3) * G::STAGE) + apiece[i] * 1024;
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
    auto loadB1 = [&](int hs, int nf, frag_t& dst) {
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
    // (two waves per SIMD: hipcc splits the 256 registers of a wave 128 / 128 between the two files)
    // (K split: none - the partial accumulators are added with ordinary VALU code, and 20-odd fragments leave room)
    constexpr int ACC_A = KS ? 0 : (PW == 8 ? 32 : 64);       // accumulator fragments kept in AGPRs
    auto mma = [&](auto IDX_, f32x4& c, const frag_t& a, const frag_t& b) {
        if constexpr ((FFM_PANEL_ABL & 4) != 0) return;
        if constexpr (decltype(IDX_)::value < ACC_A)
            asm volatile(FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else
            asm volatile(FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    };
    // RANKOP: t = A_rows . rk^T for the block's rows; wave w owns fragment rows w, w+4, ... (VGPR accumulators)
    constexpr int TI = (MF + CW - 1) / CW;
    constexpr bool WSPEC = RK && MF * NF <= 32;               // specialise the main loop per wave (t fragment rows)
    f32x4 tacc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) tacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int offA0 = frow * 128 + ((fgrp ^ (lane & 7)) << 4);
    const int offA1 = offA0 ^ 64;
    // One half-step: MF A fragments (double-buffered by hand) x NF MFMAs each.  After fragment row mf the wave issues
    // ONE vector-memory instruction (issue(mf)).
    auto half = [&](auto W_, const char* st, int off, const frag_t (&b)[NF], auto&& issue) {
        // A fragments: a ring of AD + 1 registers, AD reads in flight.  One wave per SIMD has nobody to hide an LDS
        // round trip behind: with NF MFMAs (16 cycles each) per fragment, AD * NF * 16 cycles must cover it.  The
        // reads are inline asm with counted s_waitcnt lgkmcnt (LDS returns in order): the compiler's own insertion
        // falls back to lgkmcnt(0) around the asm MFMAs and would wait for the prefetches too.
        // (two waves per SIMD: the partner covers the round trip, and the registers are needed elsewhere)
        constexpr int AD = PW == 8 ? (NF >= 3 ? 2 : 3) : (NF >= 6 ? 2 : (NF >= 4 ? 3 : 5));
        const uint32_t sa = (uint32_t)(uintptr_t)st + (uint32_t)off;
        auto lds_read = [](frag_t& dst, uint32_t addr, auto OFF_) {
            if constexpr ((FFM_PANEL_ABL & 8) != 0) return;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(OFF_)::value) : "memory");
        };
        frag_t a[AD + 1];
        frag_t kf;                                  // RANKOP: the 16 rank rows ride behind the A rows of the stage
        if constexpr (RK) lds_read(kf, sa, std::integral_constant<int, MF * 2048>{});
        static_for<(AD < MF ? AD : MF)>([&](auto I_) { lds_read(a[decltype(I_)::value], sa, std::integral_constant<int, decltype(I_)::value * 2048>{}); });
        f32x4(&tr)[TI] = tacc;                      // (named outside the if constexpr so that the lambdas capture it)
        const int wv = colw;
        static_for<MF>([&](auto MF_) {
            constexpr int mf = decltype(MF_)::value;
            if constexpr (mf + AD < MF) lds_read(a[(mf + AD) % (AD + 1)], sa, std::integral_constant<int, (mf + AD) * 2048>{});
            {
                constexpr int younger = (MF - 1 - mf) < AD ? (MF - 1 - mf) : AD;      // reads issued after a[mf]
                if constexpr (RK && mf == 0)
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(kf) : "n"(younger) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[mf % (AD + 1)]) : "n"(younger) : "memory");
            }
            static_for<NF>([&](auto NF_) {
                constexpr int nf = decltype(NF_)::value;
                mma(std::integral_constant<int, mf * NF + nf>{}, acc[mf][nf], a[mf % (AD + 1)], b[nf]);
            });
            if constexpr (RK) {
                // t fragment row mf belongs to wave mf % 4: one more MFMA on the fragment that is already in registers
                // (asm, VGPR form: a builtin MFMA would be given AGPRs and evict accumulator fragments from them).
                // The wave index is a template parameter of the loop: a run-time test here costs a taken branch per
                // fragment row, 0.4 us per K step.
                // (Tiles whose accumulators already fill the register file cannot afford four loop copies: they spill.)
                if constexpr (WSPEC) {
                    if constexpr ((mf % CW) == decltype(W_)::value)
                        asm volatile(FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+v"(tr[mf / CW]) : "v"(a[mf % (AD + 1)]), "v"(kf));
                } else {
                    if ((mf % CW) == wv)
                        asm volatile(FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+v"(tr[mf / CW]) : "v"(a[mf % (AD + 1)]), "v"(kf));
                }
            }
            issue(MF_);
        });
    };
    static_assert(NF + G::NI <= MF, "one VMEM slot per fragment row");

    // LNIN: the partial row sums of this thread's tile row are requested FIRST: behind the ring fills each of these
    // small loads waits ~150 cycles for an issue slot (3 us per block, tools/panel_stamps.py)
    constexpr bool LNIN_ = (FL & FFM_EPI_LNIN) != 0;
    // (LNB_APPLY on the plain product: up to 24 partial rows - two per head from the attention backward kernels)
    constexpr int NPV = (LNBA && !RK) ? 24 : 8;
    f32x2 lnpv[NPV];
    float lnb_mu = 0.f, lnb_rs = 0.f;
    if constexpr (LNBA) {
        // LNB_APPLY: the producer's partial row sums {P1, P2} and the LayerNorm's saved statistics of this thread's tile row
        static_assert(16 * MF <= PT, "one tile row per thread");
        const int gm = (m0 + tid) < p.M ? (m0 + tid) : (p.M - 1);
#pragma unroll
        for (int q = 0; q < NPV; ++q) {
            const int qq = q < p.lnb_np ? q : 0;
            const float* src = p.lnb_part + ((size_t)qq * p.M + gm) * 2;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lnpv[q]) : "v"(src) : "memory");
        }
        asm volatile("global_load_dword %0, %1, off" : "=v"(lnb_mu) : "v"(p.ln_mean + gm) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(lnb_rs) : "v"(p.ln_rstd + gm) : "memory");
    }
    if constexpr (LNIN_) {
        static_assert(16 * MF <= PT, "one tile row per thread");
        const int gm = (m0 + tid) < p.M ? (m0 + tid) : (p.M - 1);
        // (inline asm, consumed behind the prologue's own vmcnt(0): a compiler-visible load in front of the LDS-DMA
        // builtins is answered with s_waitcnt vmcnt(0) before the first of them, i.e. a full HBM round trip up front)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int qq = q < p.ln_np ? q : 0;                // always a valid address; surplus slots are not summed
            const float* src = p.ln_part + ((size_t)qq * p.M + gm) * 2;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lnpv[q]) : "v"(src) : "memory");
        }
    }

    // ---- persistent epilogue operands first (a handful of small loads), then the ring: see below
    // (K split: the partial accumulators of waves 4-7 are staged over the ring behind the loop, [CW][MF*NF + TI] KiB)
    constexpr int PARTB = KS ? CW * (MF * NF + (RK ? TI : 0)) * 1024 : 0;
    constexpr int PBASE = PARTB > G::RING ? PARTB : G::RING;
    float* Bias = reinterpret_cast<float*>(smem + PBASE);
    constexpr bool LNBS_ = (FL & FFM_EPI_LNB_STAT) != 0;
    bf16_t* LwB = reinterpret_cast<bf16_t*>(Bias + BNp);      // [BN][32]: LoRA matrix tile, rank slots >= r zero
    float* Sg = reinterpret_cast<float*>(LwB + BNp * 32);     // lora_S [G][r]
    float* Ssum = Sg + 256;                                   // sum_g lora_S[g][j]
    int* Ga = reinterpret_cast<int*>(Ssum + 16);              // group id of each tile row (-1: uniform mix)
    constexpr bool LNIN = (flags & FFM_EPI_LNIN) != 0, ROWST = (flags & FFM_EPI_ROWSTATS) != 0;
    float* Cv = reinterpret_cast<float*>(smem + PBASE + persist_bytes(MF, NF, RK, false, CW, false, LNBS_));     // LNIN: c [BN] (LNB_APPLY: gamma)
    float* Mu = Cv + BNp;                                     // row means [BM]
    float* Rs = Mu + BMp;                                     // row 1 / sqrt(var + eps) [BM]
    constexpr bool LNX = LNIN || LNBA;                        // (LNB_APPLY uses the LNIN block: gamma in c's place, mean, rstd)
    float* V1F = reinterpret_cast<float*>(smem + PBASE + persist_bytes(MF, NF, RK, LNX, CW, false, LNBS_));      // LGRAD: lg_v rows [BM][r]
    float* C1v = reinterpret_cast<float*>(smem + PBASE + persist_bytes(MF, NF, RK, LNX, CW, LGRAD, LNBS_));   // LNB_APPLY: c1 [BM]
    float* C2v = C1v + BMp;                                   //            c2 [BM]
    const int r = RK ? p.rank : 0;
    // bias, c (LNIN), lora_S and the rows' group ids: inline-asm loads with clamped indices, consumed behind the ONE
    // vmcnt(0) below.  As compiler-visible loads each of them (a conditional load followed by its LDS store) was answered
    // with its own s_waitcnt vmcnt(0) behind the ring fills: four to six memory round trips in a row in every launch.
    constexpr int NBI = (BNp + PT - 1) / PT, NGI = (BMp + PT - 1) / PT;
    float biasv[NBI], cvv[NBI], sgv = 0.f;
    int gav[NGI];
    auto ldgf = [](const float* q) -> float {
        float v;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
        return v;
    };
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
        const int i = tid + it * PT, ic = i < BNp ? i : BNp - 1;
        biasv[it] = cvv[it] = 0.f;
        if constexpr ((flags & FFM_EPI_BIAS) != 0) biasv[it] = ldgf(p.bias + n0 + ic);
        if constexpr (LNIN) cvv[it] = ldgf(p.ln_c + n0 + ic);
        if constexpr (LNBA) cvv[it] = ldgf(p.lnb_gamma + n0 + ic);
    }
#pragma unroll
    for (int it = 0; it < NGI; ++it) gav[it] = -1;
    if constexpr (RK) {
        const int gr = p.G * r;
        sgv = ldgf(p.S + (tid < gr ? tid : gr - 1));
        if (p.attr) {
#pragma unroll
            for (int it = 0; it < NGI; ++it) {
                const int i = tid + it * PT, ic = i < BMp ? i : BMp - 1;
                const int gm = (m0 + ic) < p.M ? (m0 + ic) : (p.M - 1);
                asm volatile("global_load_dword %0, %1, off" : "=v"(gav[it]) : "v"(p.attr + gm / p.rows_per_sample) : "memory");
            }
        }
    }
    if constexpr (RK) {
        if (p.lw_wide) {
            // the tile [BN][32] of the pre-packed LoRA matrix (ffm_lora_pack_multi, dst_wide) is one contiguous
            // BN * 64 bytes: straight into LDS by DMA, 1 KiB pieces dealt over the waves (no VALU, no LDS stores; the
            // conversion loop below cost 3-6.5 us of VMEM issue behind the ring fills, tools/panel_stamps.py)
            const char* wsrc = reinterpret_cast<const char*>(p.lw_wide) + (size_t)n0 * 64 + lane * 16;
            for (int q = wave; q < BNp / 16; q += PW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + q * 1024),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(LwB) + q * 1024),
                                                 16, 0, 0);
        } else {
            // LoRA matrix tile: thread -> column n; all of its global loads are issued before the first LDS store
            // (a load -> store loop pays one memory round trip per iteration), then one 64-byte row is written.
            constexpr int NCOL = (BNp + PT - 1) / PT;
            float lwv[NCOL][16];
    #pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                const int n = tid + c * PT;
    #pragma unroll
                for (int j = 0; j < 16; ++j) {
                    lwv[c][j] = 0.f;
                    if (j < r && n < BNp)
                        lwv[c][j] = (flags & FFM_EPI_LORA_KR) ? p.lw[(size_t)(n0 + n) * r + j] : p.lw[(size_t)j * p.N + n0 + n];
                }
            }
    #pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                const int n = tid + c * PT;
                if (n < BNp) {
                    bf16x8 lo8, hi8, z8;
    #pragma unroll
                    for (int j = 0; j < 8; ++j) { lo8[j] = (bf16_t)lwv[c][j]; hi8[j] = (bf16_t)lwv[c][8 + j]; z8[j] = (bf16_t)0.f; }
                    bf16x8* row = reinterpret_cast<bf16x8*>(LwB + n * 32);
                    row[0] = lo8; row[1] = hi8; row[2] = z8; row[3] = z8;
                }
            }
        }
    }
    if constexpr (LGRAD) {
        // the tile's BM * r floats of lg_v ([M][r] fp32) are contiguous: 1 KiB pieces dealt over the waves, chunks beyond
        // the last row clamped (their rows are masked where they are used)
        const long lim = (long)p.M * r - 4;
        for (int q = wave; q * 256 < BMp * r; q += PW) {
            long e0 = (long)m0 * r + q * 256 + lane * 4;
            e0 = e0 < lim ? e0 : lim;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lg_v + e0),
                                             (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(V1F) + q * 1024),
                                             16, 0, 0);
        }
    }
    FFM_STAMP(6);
    // ---- prologue of the ring: stages 0..2 and the B fragments of half-steps 0..2 in the order the steady state would
    // have issued them (step k issues Bh(2k+3), A(k+3), Bh(2k+4); run backwards from k = 0 that is A0, A1, Bh0, Bh1, A2,
    // Bh2), so that the loop's own counted waits are exact from its first step on and the block only has to wait HERE
    // for stage 0 and the operands above - everything issued before A1 - not for all three stages (round 3; the full
    // drain cost ~1 us of every launch, tools/panel_stamps.py "pro: wait for all")
#if defined(FFM_PANEL_FULL_DRAIN)
    dma(0); dma(1); dma(2);
    fence();
    loadB(0, bq[0]); loadB(1, bq[1]); loadB(2, bq[2]);
    fence();
    FFM_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if constexpr (KS) {
        // K split: a wave loads the B fragments of ITS half of a step only; steady-state order per step is B(k), A(k)
        loadB(0 + kg, bq[0]);
        fence();
        dma(0);
        fence();
        loadB(2 + kg, bq[1]);
        fence();
        dma(1);
        fence();
        loadB(4 + kg, bq[2]);
        fence();
        dma(2);
        fence();
        FFM_STAMP(7);
        wait_vm<2 * G::NI + 2 * NF>();                          // everything up to and including stage 0
    } else {
        dma(0);
        dma(1);
        fence();
        loadB(0, bq[0]);
        loadB(1, bq[1]);
        fence();
        dma(2);
        fence();
        loadB(2, bq[2]);
        fence();
        FFM_STAMP(7);
        wait_vm<2 * G::NI + 3 * NF>();
    }
#endif
    asm volatile("" : "+v"(sgv));
#pragma unroll
    for (int it = 0; it < NBI; ++it) {
        asm volatile("" : "+v"(biasv[it]), "+v"(cvv[it]));
        const int i = tid + it * PT;
        if (i < BNp) {
            Bias[i] = biasv[it];
            if constexpr (LNX) Cv[i] = cvv[it];
        }
    }
    if constexpr (RK) {
        if (tid < p.G * r) Sg[tid] = sgv;
#pragma unroll
        for (int it = 0; it < NGI; ++it) {
            asm volatile("" : "+v"(gav[it]));
            const int i = tid + it * PT;
            if (i < BMp) Ga[i] = gav[it];
        }
    }
    if constexpr (LNIN) {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(lnpv[q]));
        if (tid < BMp) {
            const int i = tid;
            const int gm = (m0 + i) < p.M ? (m0 + i) : (p.M - 1);
            float su = 0.f, sq = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {                  // fixed order over the partials
                if (q < p.ln_np) {
                    su += lnpv[q][0];
                    sq += lnpv[q][1];
                }
            }
            const float mu = su / (float)p.K;
            float var = sq / (float)p.K - mu * mu;
            var = var > 0.f ? var : 0.f;
            const float rs = 1.0f / sqrtf(var + 1e-5f);
            Mu[i] = mu;
            Rs[i] = rs;
            if (tn == 0 && m0 + i < p.M) {
                if (p.ln_mean) p.ln_mean[gm] = mu;
                if (p.ln_rstd) p.ln_rstd[gm] = rs;
            }
        }
    }
    if constexpr (LNBA) {
#pragma unroll
        for (int q = 0; q < NPV; ++q) asm volatile("" : "+v"(lnpv[q]));
        asm volatile("" : "+v"(lnb_mu), "+v"(lnb_rs));
        if (tid < BMp) {
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int q = 0; q < NPV; ++q) {                // fixed order over the producer's column tiles
                if (q < p.lnb_np) {
                    p1 += lnpv[q][0];
                    p2 += lnpv[q][1];
                }
            }
            Mu[tid] = lnb_mu;
            Rs[tid] = lnb_rs;
            C1v[tid] = p1;                                 // (the rank-r corrections are added behind the main loop)
            C2v[tid] = p2;
        }
    }
    __syncthreads();
    if constexpr (LGRAD) {
        // lg_v tile -> hi | lo pairs in place, rows beyond M zeroed (read again in the epilogue, many barriers from here)
        for (int e = tid; e < BMp * r; e += PT)
            reinterpret_cast<uint32_t*>(V1F)[e] = (m0 + e / r < p.M) ? lg_split(V1F[e]) : 0u;
    }
    FFM_STAMP(1);

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
    auto step = [&](auto W_, int kt, auto P_, auto TAIL_) {
        constexpr int P = decltype(P_)::value;
        constexpr bool TAIL = decltype(TAIL_)::value != 0;      // tail: the last four steps, guarded by rem
        const int rem = KT - kt;
        const char* st = smem + (kt & 3) * G::STAGE;
        if (!TAIL || rem >= 3) wait_vm<2 * NF + nA>();
        else if (rem == 2) wait_vm<2 * NF>();
        else wait_vm<NF>();
        tieB(bq[2 * P]);
        half(W_, st, offA0, bq[2 * P], [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < NF) {
                if (!(FFM_PANEL_ABL & 1) && (!TAIL || rem >= 2)) loadB1(2 * kt + 3, j, bq[(2 * P + 3) & 3][j]);
            } else if constexpr (j < NF + nA) {
                if (!(FFM_PANEL_ABL & 2) && (!TAIL || rem >= 4)) dma_piece(kt + 3, j - NF);
            }
        });
        fence();
        if (!TAIL || rem >= 4) wait_vm<2 * NF + 2 * nA>();
        else if (rem == 3) wait_vm<2 * NF + nA>();
        else if (rem == 2) wait_vm<2 * NF>();
        else wait_vm<0>();
        tieB(bq[2 * P + 1]);
        half(W_, st, offA1, bq[2 * P + 1], [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < NF) {
                if (!(FFM_PANEL_ABL & 1) && (!TAIL || rem >= 3)) loadB1(2 * kt + 4, j, bq[2 * P][j]);
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
    // K split: ONE half per step and wave (its K half of the stage), the B ring indexed by step (slot = kt & 3, loaded
    // three steps ahead).  VMEM order of step kt on every wave: B(kt+3) = NF fragments, then the nA pieces of A(kt+3).
    //   start of step kt needs B(kt):   younger = A(kt) + steps kt+1, kt+2            = 2 NF + 3 nA   (rem 2: NF + 2 nA, rem 1: nA)
    //   end of step kt needs A(kt+1):   younger = steps kt+2, kt+3                     = 2 NF + 2 nA   (rem 3: NF + nA, rem 2: 0)
    auto step_ks = [&](auto W_, int kt, auto S_, auto TAIL_) {
        constexpr int S = decltype(S_)::value;
        constexpr bool TAIL = decltype(TAIL_)::value != 0;
        const int rem = KT - kt;
        const char* st = smem + (kt & 3) * G::STAGE;
        if (!TAIL || rem >= 3) wait_vm<2 * NF + 3 * nA>();
        else if (rem == 2) wait_vm<NF + 2 * nA>();
        else wait_vm<nA>();
        tieB(bq[S]);
        half(W_, st, kg ? offA1 : offA0, bq[S], [&](auto J_) {
            constexpr int j = decltype(J_)::value;
            if constexpr (j < NF) {
                if (!TAIL || rem >= 4) loadB1(2 * (kt + 3) + kg, j, bq[(S + 3) & 3][j]);
            } else if constexpr (j < NF + nA) {
                if (!TAIL || rem >= 4) dma_piece(kt + 3, j - NF);
            }
        });
        fence();
        if (!TAIL || rem >= 2) {
            if (!TAIL || rem >= 4) wait_vm<2 * NF + 2 * nA>();
            else if (rem == 3) wait_vm<NF + nA>();
            else wait_vm<0>();
            block_barrier();
        }
    };
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto main_loop = [&](auto W_) {
        int kt = 0;
        if constexpr (KS) {
            for (; kt < KT - 4; kt += 4) {                       // (the selector admits K % 256 == 0 only: KT % 4 == 0)
                step_ks(W_, kt, I0{}, I0{});
                step_ks(W_, kt + 1, I1{}, I0{});
                step_ks(W_, kt + 2, I2{}, I0{});
                step_ks(W_, kt + 3, I3{}, I0{});
            }
            step_ks(W_, kt, I0{}, I1{});
            step_ks(W_, kt + 1, I1{}, I1{});
            step_ks(W_, kt + 2, I2{}, I1{});
            step_ks(W_, kt + 3, I3{}, I1{});
        } else {
            for (; kt < KT - 4; kt += 2) {                       // steady state: no guards, no branches
                step(W_, kt, I0{}, I0{});
                step(W_, kt + 1, I1{}, I0{});
            }
            for (; kt < KT; kt += 2) {                           // last four steps
                step(W_, kt, I0{}, I1{});
                step(W_, kt + 1, I1{}, I1{});
            }
        }
    };
    // Two waves per SIMD: the second-dispatched half (waves 4-7) loses every issue arbitration to its older partner (priority,
    // then age: MI355X_MICROARCH.md, "Two waves per SIMD", item 4).  ONE static s_setprio 1 for that half in front of the
    // loop, no per-segment flips; back to 0 for the epilogue.  Measured (round 4, A/B in one gpurun call, three alternating pairs of
    // 40-step runs): 4.9376 / 4.9423 / 4.9213 ms per step with it, 4.9328 / 4.9497 / 4.9245 without - nothing; off by default
    // (-DFFM_PANEL_PRIO=1 builds it).
#ifndef FFM_PANEL_PRIO
#define FFM_PANEL_PRIO 0
#endif
    if constexpr (PW == 8 && !KS && FFM_PANEL_PRIO) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    if constexpr (WSPEC) {                                       // one copy of the loop per column slab (see half())
        static_for<CW>([&](auto WW_) {
            if (colw == decltype(WW_)::value) main_loop(WW_);
        });
    } else {
        main_loop(I0{});
    }
    if constexpr (PW == 8 && !KS && FFM_PANEL_PRIO) __builtin_amdgcn_s_setprio(0);
    // the asm MFMAs are invisible to the hazard recogniser: let the last ones retire before the accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();                                                  // the ring is free from here on
    constexpr bool ROWST_ = (FL & FFM_EPI_ROWSTATS) != 0;
    if constexpr (KS) {
        // waves 4-7 hand their partial accumulators (and partial t) to the wave of the same column slab through LDS,
        // then only keep the remaining barriers of the block company: the epilogue belongs to waves 0-3
        f32x4* Pacc = reinterpret_cast<f32x4*>(smem) + ((size_t)colw * (MF * NF + (RK ? TI : 0))) * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) Pacc[(i * NF + j) * 64] = acc[i][j];
            if constexpr (RK) {
#pragma unroll
                for (int i = 0; i < TI; ++i) Pacc[(MF * NF + i) * 64] = tacc[i];
            }
        }
        __syncthreads();                                              // partials visible
        if (kg == 1) {
            __syncthreads();                                          // (partials consumed)
            if constexpr (RK) __syncthreads();                        // (the rank-r stage's barrier)
            if constexpr (ROWST_) __syncthreads();                    // (the row sums' barrier)
            return;
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const f32x4 o = Pacc[(i * NF + j) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] += o[e];
            }
        if constexpr (RK) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const f32x4 o = Pacc[(MF * NF + i) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) tacc[i][e] += o[e];
            }
        }
        __syncthreads();                                              // partials consumed: the staging area is free
    }
    constexpr int ET = CW * 64;                                       // threads that run the epilogue
    FFM_STAMP(2);

    // ---------------- epilogue ----------------
    constexpr int PITCH = stage_pitch(NF);
    constexpr int CPR = 2 * NF;                       // 8-column chunks per row of the wave's slab
    constexpr int NRG = (MF + 1) / 2;                 // 32-row groups
    // row groups of residual / pre-activation rows kept in flight (two waves per SIMD: the register budget is 256)
    constexpr int PFMAX = (PW == 8 && NF >= 3) ? 2 : 3;
    // (measured in round 6 and not kept: ALL five row groups of the LNB_APPLY tiles' two operand streams requested up front -
    // 20 loads per lane in flight, no counted wait behind the first - 4.517 against 4.500 ms per step for the three-deep version,
    // profiles/r06_pfall_ab.txt)
    constexpr int PF = NRG < PFMAX ? NRG : PFMAX;
    // RANKOP: per-wave dS sums at smem + 0 (DsP below), then
    bf16_t* TsA = reinterpret_cast<bf16_t*>(smem + BMp * 64);         // ts tile [BM][32] bf16, zero padded
    uint32_t* V2F = reinterpret_cast<uint32_t*>(smem + BMp * 128);    // LGRAD: the same rows as fp32-accurate hi | lo pairs, [BM][16]
    float* Cw = reinterpret_cast<float*>(smem + (RK ? BMp * 192 : 0)) + colw * (32 * PITCH);
    // ROWSTATS: per-wave partial row sums [PW][BM][2] behind the four waves' output stages
    float* RowP = reinterpret_cast<float*>(smem + (RK ? BMp * 192 : 0)) + CW * (32 * PITCH);
    static_assert((CPR & (CPR - 1)) == 0 || !(FL & FFM_EPI_ROWSTATS), "row sums: the lanes of a row form a power-of-two group");
    bf16_t* C = reinterpret_cast<bf16_t*>(p.c);

    // residual / pre-activation rows of the first row groups: issued now, consumed after the rank-r update
    constexpr bool PRE = (flags & (FFM_EPI_RESIDUAL | FFM_EPI_DGELU)) != 0 || LNBA;
    const bf16_t* prep = reinterpret_cast<const bf16_t*>(((flags & FFM_EPI_RESIDUAL) || LNBA) ? p.res : p.aux);
    bf16x8 rpre[PF][NF];
    // LNB_APPLY: a second stream beside the residual gradient - the LayerNorm's input rows; the two loads of a chunk are
    // issued back to back (LPB loads per chunk in the counted waits below)
    constexpr int LPB = LNBA ? 2 : 1;
    const bf16_t* prep2 = reinterpret_cast<const bf16_t*>(p.lnb_x);
    bf16x8 rpre2[LNBA ? PF : 1][NF];
    // The loads are inline asm with hand-counted waits, like the weight fragments of the main loop: left to the
    // compiler, every use of a prefetched row group became s_waitcnt vmcnt(0), which also waits for the stores just
    // issued and for the two younger prefetches (dX(c_proj): 26 us of output epilogue for 13 us of HBM traffic).
    // Rows beyond M (last row tile) are clamped: loaded, never stored.
    auto load_pre = [&](int rg, bf16x8 (&dst)[NF], bf16x8 (&dst2)[NF]) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int idx = lane + 64 * i, row = idx / CPR, ch = idx % CPR;
            int gm = m0 + rg * 32 + row;
            gm = gm < p.M ? gm : p.M - 1;
            const bf16_t* src = prep + (size_t)gm * p.ldc + n0w + ch * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[i]) : "v"(src) : "memory");
            if constexpr (LNBA) {
                const bf16_t* src2 = prep2 + (size_t)gm * p.ldc + n0w + ch * 8;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst2[i]) : "v"(src2) : "memory");
            }
        }
    };
    // VMEM operations younger than the load of (row group q >= PF, chunk i), which is issued at the end of group q - PF:
    // the rest of its batch, groups q-PF+1 .. q-1 (NF * SPC stores each, plus a batch of NF loads while one is left to
    // issue), and this group's stores so far.  Exact when every lane stores (full tiles); the last group of an odd MF
    // has chunks without rows, so its own stores are not counted (waits a little longer than needed).
    constexpr int SPC = (flags & FFM_EPI_GELU) ? 2 : 1;
    const bool full_tile = m0 + BMp <= p.M;
    if constexpr (PRE) {
#pragma unroll
        for (int g = 0; g < PF; ++g) load_pre(g, rpre[g], rpre2[LNBA ? g : 0]);
    }

    if constexpr (RK) {
        // pi_b[g] = lambda on the sample's own group, (1 - lambda) / (G - 1) elsewhere, 1 / G without an attribute:
        //   s_b[j] = sum_g pi_b[g] S[g][j] = w_o * ssum[j] + (lambda - w_o) * S[a][j]      (a >= 0), ssum[j] = sum_g S[g][j]
        // (one LDS read and one FMA per entry instead of a loop over the groups with a division in it)
        const float w_own = p.lambda_group, w_oth = (1.0f - p.lambda_group) / (float)(p.G > 1 ? p.G - 1 : 1);
        const float w_uni = 1.0f / (float)p.G;
        // Every block of a tile row holds the same t: the stores of t / ts and the dS partial sums are split over
        // the tiles_n blocks by FRAGMENT ROWS (block tn takes the fragment rows mf = tn mod tiles_n), so that no block
        // becomes a straggler of this single-round kernel and the choice is a wave-uniform branch.
        const bool do_ds = p.t_fwd && p.ds_part;
        // t stays in the registers of the wave whose MFMAs made it (fragment row mfi belongs to wave mfi % 4; lane =
        // (column j = frow, rows 4 * fgrp + e)): ts, the t / ts stores and the dS products are formed right there, only
        // the bf16 ts tile goes through LDS (it is the A operand of the rank-r update of ALL four waves).  Before, t
        // made a round trip through LDS into a thread-per-element loop with the t_fwd loads inside it, and 24 threads
        // summed the dS products row by row: 4-10 us per launch (tools/panel_stamps.py), now one barrier.
        const int j = frow;
        const bool jok = j < r;
        // LNIN: rk holds (gamma (.) lora_A)^T, so t = rstd (acc - mu c_j) + d_j = LayerNorm(x) lora_A
        float lncj = 0.f, lndj = 0.f;
        if constexpr (LNIN) {
            lncj = p.ln_rk[j];
            lndj = jok ? p.ln_rk[16 + j] : 0.f;
        }
        if constexpr (LNBA) {                          // (A^T gamma)[j], (A^T beta)[j]
            lncj = jok ? p.ln_rk[j] : 0.f;
            lndj = jok ? p.ln_rk[16 + j] : 0.f;
        }
        float ssum = 0.f;
#pragma unroll
        for (int g = 0; g < FFM_MAX_GROUPS; ++g)
            if (g < p.G && jok) ssum += Sg[g * r + j];
        float d_all = 0.f, d_uni = 0.f, d_own[FFM_MAX_GROUPS];
#pragma unroll
        for (int g = 0; g < FFM_MAX_GROUPS; ++g) d_own[g] = 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int mfi = colw + CW * i;
            if (mfi < MF) {
                const bool mine = __builtin_amdgcn_readfirstlane((int)((unsigned)mfi % (unsigned)tiles_n)) == tn;
                const int row0 = mfi * 16 + fgrp * 4;
                // (three passes, so that the LDS / global reads of the four elements are in flight together)
                int ga[4];
                float sgv[4], tfw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ga[e] = Ga[row0 + e];
                    tfw[e] = 0.f;
                }
                if (mine && do_ds && jok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m0 + row0 + e < p.M) tfw[e] = p.t_fwd[(size_t)(m0 + row0 + e) * r + j];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) sgv[e] = (jok && ga[e] >= 0) ? Sg[ga[e] * r + j] : 0.f;
                float tsv[4];
                if constexpr (LNIN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) tacc[i][e] = Rs[row0 + e] * (tacc[i][e] - Mu[row0 + e] * lncj) + lndj;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sb = ga[e] < 0 ? w_uni * ssum : w_oth * ssum + (w_own - w_oth) * sgv[e];
                    tsv[e] = (jok && m0 + row0 + e < p.M) ? p.scaling * tacc[i][e] * sb : 0.f;
                    // LNIN: the output epilogue multiplies the WHOLE accumulator by rstd (rstd (x W'^T - mu c) + d), so
                    // the rank-r term enters divided by it: rstd (x W'^T - mu c + (ts / rstd) lw) + d
                    if constexpr (LNIN) TsA[(row0 + e) * 32 + j] = (bf16_t)(tsv[e] / Rs[row0 + e]);
                    else
                    TsA[(row0 + e) * 32 + j] = (bf16_t)tsv[e];
                    TsA[(row0 + e) * 32 + 16 + j] = (bf16_t)0.f;
                    if constexpr (LGRAD) V2F[(row0 + e) * 16 + j] = lg_split(tsv[e]);      // ts rows as hi | lo pairs (zero beyond M and beyond r)
                }
                if constexpr (LNBA) {
                    // the rank-r part of the two LayerNorm-backward row sums: c1 += sum_j us[j] (A^T gamma)[j],
                    // c2 -= sum_j us[j] (A^T beta)[j] - tsv IS us (rank slot j = frow of row 4 fgrp + e; zero beyond r and M);
                    // the 16 slots of a row sit in one 16-lane group: four butterfly steps, fixed order
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (rank slots 14 / 15 of the operand hold W gamma and d: the raw t there IS sum_n dpre (W gamma) / sum_n dpre d)
                        float k1 = tsv[e] * lncj + (j == 14 ? tacc[i][e] : 0.f), k2 = tsv[e] * lndj + (j == 15 ? tacc[i][e] : 0.f);
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) {
                            k1 += __shfl_xor(k1, o, 64);
                            k2 += __shfl_xor(k2, o, 64);
                        }
                        if (frow == 0) {
                            C1v[row0 + e] += k1;
                            C2v[row0 + e] -= k2;
                        }
                    }
                }
                if (mine && jok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int gm = m0 + row0 + e;
                        if (gm < p.M) {
                            if (p.t_out) p.t_out[(size_t)gm * r + j] = tacc[i][e];
                            if (p.ts_out) p.ts_out[(size_t)gm * r + j] = tsv[e];
                        }
                        const float wv = p.scaling * tfw[e] * tacc[i][e];      // tfw = 0 beyond M and without dS
                        if (ga[e] < 0) d_uni += wv; else d_all += wv;
#pragma unroll
                        for (int g = 0; g < FFM_MAX_GROUPS; ++g) d_own[g] += (ga[e] == g) ? wv : 0.f;
                    }
                }
            }
        }
        FFM_STAMP(8);
        float* DsP = reinterpret_cast<float*>(smem);                  // [PW][FFM_MAX_GROUPS + 2][16]: per-wave dS sums
        if (do_ds) {
            // dS partial of this block's row slice: sum_rows pi_b[g] * scaling * t_fwd * t
            //   = w_o * sum_rows wv + (lambda - w_o) * sum_{rows of group g} wv     (uniform mix: w_uni * sum_rows wv)
            // fixed order: the lane's rows, the four row groups of the wave (two butterfly steps), then the waves
            auto rows4 = [](float v) {
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                return v;
            };
            d_all = rows4(d_all);
            d_uni = rows4(d_uni);
#pragma unroll
            for (int g = 0; g < FFM_MAX_GROUPS; ++g)
                if (g < p.G) d_own[g] = rows4(d_own[g]);
            if (fgrp == 0) {
                float* dst = DsP + colw * ((FFM_MAX_GROUPS + 2) * 16) + j;
                dst[0] = d_all;
                dst[16] = d_uni;
#pragma unroll
                for (int g = 0; g < FFM_MAX_GROUPS; ++g)
                    if (g < p.G) dst[(2 + g) * 16] = d_own[g];
            }
        }
        __syncthreads();
        FFM_STAMP(9);
        if (do_ds && tid < p.G * r) {
            const int g = tid / r, jj = tid % r;
            float all = 0.f, own = 0.f, uni = 0.f;
#pragma unroll
            for (int w = 0; w < CW; ++w) {
                const float* src = DsP + w * ((FFM_MAX_GROUPS + 2) * 16) + jj;
                all += src[0];
                uni += src[16];
                own += src[(2 + g) * 16];
            }
            p.ds_part[((size_t)(tm * tiles_n + tn) * p.G + g) * r + jj] = w_uni * uni + w_oth * all + (w_own - w_oth) * own;
        }
        // rank-r update on the matrix cores: acc += TsA . LwB^T (K = 32 rank slots, zero padded)
        frag_t lb[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            lb[nf] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(LwB) + (colw * WN + nf * 16 + frow) * 64 + fgrp * 16);
        FFM_STAMP(10);
        // ts fragments in batches ahead of their MFMAs (the weight-fragment ring is dead; a tile whose accumulators
        // and pre-activation rows already fill the register file takes smaller batches)
        constexpr int TB = (MF * NF > 64 || (PW == 8 && NF >= 3)) ? 4 : MF;
        static_for<(MF + TB - 1) / TB>([&](auto B_) {
            constexpr int b0 = decltype(B_)::value * TB;
            constexpr int nb = (MF - b0) < TB ? (MF - b0) : TB;
            frag_t ta[nb];
#pragma unroll
            for (int q = 0; q < nb; ++q)
                ta[q] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(TsA) + ((b0 + q) * 16 + frow) * 64 + fgrp * 16);
            const f32x4(&accr)[MF][NF] = acc;
            (void)accr;
            static_for<nb>([&](auto Q_) {
                constexpr int mf = b0 + decltype(Q_)::value;
                static_for<NF>([&](auto NF_) {
                    constexpr int nf = decltype(NF_)::value;
                    mma(std::integral_constant<int, mf * NF + nf>{}, acc[mf][nf], ta[decltype(Q_)::value], lb[nf]);
                });
            });
            fence();
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }

    FFM_STAMP(3);
    // per wave: 32-row groups of the accumulator slab through a private LDS stage, then 16-byte row segments out.
    // accumulator fragment -> stage: explicit ds_write_b32 (data straight from the AGPR / VGPR the MFMAs left it in)
    const uint32_t cw_lane = (uint32_t)(uintptr_t)(Cw + fgrp * 4 * PITCH + frow);
    // LGRAD.  part_c[tm][n][j] = sum_rows c[row][n] lg_v[row][j], part_a[tm][n][j] = sum_rows quick_gelu(aux)[row][n] ts[row][j]
    // over the block's rows: products V^T [16 x 32 rows] . X [32 rows x 16 columns] per row group, with the ROW index
    // contracted.  X is needed with 8 consecutive rows of one column per lane, and the epilogue has 8 consecutive columns
    // of one row per lane: every lane drops its 16-bit chunk (the rows as stored / the activation recomputed from the
    // pre-activation chunk it holds anyway) into a [32][WN] image over the part of the wave's stage that has been read
    // already, and ds_read_b64_tr_b16 hands the image back transposed (the reduction kernel's own recipe, lora.hip).
    // V enters as a bf16 hi + lo pair (fp32-level accuracy in v, as there).  No VMEM in here: the counted waits stand.
    // Both images are written INSIDE the chunk loop (the activation shares the sigmoid of the derivative there: recomputed
    // behind the loop its exp + rcp cost 12 us per launch), interleaved row by row - image row R of both at 2 TROW R, 16
    // bytes short of the stage's row pitch - so that what iteration i writes (rows it has just read) ends below the first
    // stage row a LATER iteration still reads, except in a row the two iterations share: there it covers the row's first
    // 4 PITCH - 16 (R + 1) bytes, chunks the earlier iteration has consumed (NF = 3: row 10, its chunk 0; row 21: nothing).
    constexpr int TROW = WN * 2;                                      // bytes per image row
    static_assert(!LGRAD || (NF == 3 && 2 * 32 * TROW <= 32 * PITCH * 4), "both images fit the wave's stage; shared rows checked for NF = 3");
    // (the accumulators are born in the first row group, behind its stage writes: the AGPRs of the main accumulator's
    // first two fragment rows are free from there on, and the epilogue has no VGPRs to spare on the 8-wave tile)
    f32x4 lgc[NF], lga[NF];
    // LNB_STAT: per chunk {s1, s2} of the row group in flight, [32 rows][CPR] f32x2 in the wave's own slab of the LoRA-matrix
    // tile (dead once `lb` above has been read: wave-private, no barrier); per wave and row the sums go to RowP
    const uint32_t lnq = (uint32_t)(uintptr_t)(reinterpret_cast<char*>(LwB) + colw * WN * 64);
    static_assert(!LNBS || 32 * CPR * 8 <= WN * 64, "the chunk sums of a row group fit the wave's LoRA-matrix slab");
    const uint32_t t_img = (uint32_t)(uintptr_t)Cw;
    const uint32_t t_rd = t_img + (8 * fgrp + (frow >> 2)) * 2 * TROW + 8 * (lane & 3);     // lane 4q + p of a group: row q, columns 4p..
    static_for<NRG>([&](auto RG_) {
        constexpr int rg = decltype(RG_)::value;
        static_for<2 * NF * 4>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            constexpr int q = t / (NF * 4), nf = (t / 4) % NF, e = t % 4, mf = 2 * rg + q;
            constexpr int off = ((q * 16 + e) * PITCH + nf * 16) * 4;
            const uint32_t cwl = cw_lane;               // (named outside the if constexpr so that the lambda captures them)
            const f32x4(&accr)[MF][NF] = acc;
            if constexpr (mf < MF) {
                if constexpr (mf * NF + nf < ACC_A)
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
            bf16x8 d8s;                                      // LNB_STAT: the chunk as stored (dpre)
            (void)d8s;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(&Cw[row * PITCH + ch * 8]);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(&Cw[row * PITCH + ch * 8 + 4]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bias[colw * WN + ch * 8]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bias[colw * WN + ch * 8 + 4]);
            if constexpr (LNIN) {
                // rstd (x W'^T - mu c) + d: the rows went through the matrix cores un-normalised
                const int trow = rg * 32 + row < BMp ? rg * 32 + row : BMp - 1;
                const float mu = Mu[trow], rs = Rs[trow];
                const f32x4 cv0 = *reinterpret_cast<const f32x4*>(&Cv[colw * WN + ch * 8]);
                const f32x4 cv1 = *reinterpret_cast<const f32x4*>(&Cv[colw * WN + ch * 8 + 4]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = rs * (c0[c] - mu * cv0[c]) + b0[c];
                    v[4 + c] = rs * (c1[c] - mu * cv1[c]) + b1[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[c] = c0[c] + b0[c]; v[4 + c] = c1[c] + b1[c]; }
            }
            if constexpr (PRE) {
                if constexpr (rg < PF) {
                    // issued before the rank-r update: one drain in front of the first use covers the first PF groups
                    if (rg == 0 && i == 0) wait_vm<0>();
                } else {
                    constexpr int last_odd = (rg == NRG - 1 && (MF & 1)) ? 1 : 0;
                    constexpr int batches = (PF - 1) - ((rg - 1 + PF >= NRG) ? 1 : 0) - ((rg - 2 + PF >= NRG && PF >= 3) ? 1 : 0);
                    static_assert(PF <= 3, "the count below walks at most two groups back");
                    constexpr int young_base = (PF - 1) * NF * SPC + (batches > 0 ? batches : 0) * NF * LPB;
                    // (i is a loop variable of an unrolled loop: the switch folds to one immediate per copy)
                    const int young = young_base + (NF - 1 - i) * LPB + (last_odd ? 0 : i * SPC);
                    if (!full_tile) wait_vm<0>();
                    else switch (young) {
#define FFM_WV(n) case n: wait_vm<n>(); break;
                        FFM_WV(0) FFM_WV(1) FFM_WV(2) FFM_WV(3) FFM_WV(4) FFM_WV(5) FFM_WV(6) FFM_WV(7) FFM_WV(8) FFM_WV(9)
                        FFM_WV(10) FFM_WV(11) FFM_WV(12) FFM_WV(13) FFM_WV(14) FFM_WV(15) FFM_WV(16) FFM_WV(17) FFM_WV(18)
                        FFM_WV(19) FFM_WV(20) FFM_WV(21) FFM_WV(22) FFM_WV(23) FFM_WV(24) FFM_WV(25) FFM_WV(26) FFM_WV(27)
                        FFM_WV(28) FFM_WV(29) FFM_WV(30) FFM_WV(31) FFM_WV(32) FFM_WV(33) FFM_WV(34) FFM_WV(35) FFM_WV(36)
                        FFM_WV(37) FFM_WV(38) FFM_WV(39) FFM_WV(40) FFM_WV(41) FFM_WV(42) FFM_WV(43) FFM_WV(44) FFM_WV(45)
                        FFM_WV(46) FFM_WV(47) FFM_WV(48)
#undef FFM_WV
                        default: wait_vm<0>();
                    }
                }
                asm volatile("" : "+v"(rpre[rg % PF][i]));
                if constexpr (LNBA) asm volatile("" : "+v"(rpre2[rg % PF][i]));
            }
            if constexpr ((flags & FFM_EPI_RESIDUAL) != 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] += (float)rpre[rg % PF][i][c];
            }
            if constexpr (LNBA) {
                // LayerNorm backward on the row: rstd (gamma g_h - c1/K - xhat c2/K) + the gradient of the residual path
                const int trow = rg * 32 + row < BMp ? rg * 32 + row : BMp - 1;
                const float mu = Mu[trow], rs = Rs[trow], invk = 1.0f / (float)p.N;
                const float c1 = C1v[trow] * invk, c2 = C2v[trow] * invk;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(&Cv[colw * WN + ch * 8]);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(&Cv[colw * WN + ch * 8 + 4]);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float xh = ((float)rpre2[rg % PF][i][c] - mu) * rs;
                    const float gm_ = c < 4 ? g0[c & 3] : g1[c & 3];
                    v[c] = rs * (gm_ * v[c] - c1 - xh * c2) + (float)rpre[rg % PF][i][c];
                }
            }
            if constexpr (LGRAD) {
                // (gelu_deriv == 0 here: the launcher checks) derivative and activation from one sigmoid; both chunks into
                // their images (asm: ordered behind this iteration's reads of the stage, whatever the compiler thinks of
                // the types)
                bf16x8 d8, a8;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float ga, gd;
                    Act<bf16_t>::gelu_both((float)rpre[rg % PF][i][c], ga, gd);
                    v[c] *= gd;
                    d8[c] = (bf16_t)v[c];
                    a8[c] = (bf16_t)ga;
                }
                if constexpr (LNBS) d8s = d8;
                if constexpr (rg == NRG - 1 && (MF & 1)) {
                    // the last group of an odd MF has 16 rows; the other 16 stage rows hold what the previous group's images
                    // left there - any bit pattern, NaN included, and 0 x NaN is what a masked V row would make of it
                    if (row >= 16) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) d8[c] = a8[c] = (bf16_t)0.f;
                    }
                }
                const uint32_t ta = t_img + row * (2 * TROW) + ch * 16;
                asm volatile("ds_write_b128 %0, %1" ::"v"(ta), "v"(d8) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ta), "v"(a8), "n"(TROW) : "memory");
            } else if constexpr ((flags & FFM_EPI_DGELU) != 0) {
                if (p.gelu_deriv) {                              // (kernel argument: a scalar branch) aux holds gelu'(pre)
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] *= (float)rpre[rg % PF][i][c];
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] *= Act<bf16_t>::gelu_grad((float)rpre[rg % PF][i][c]);
                }
            }
            if constexpr (ROWST) {
                // partial LayerNorm sums of the row AS STORED (bf16): the lanes of a row are adjacent (CPR of them)
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = (float)(bf16_t)v[c];
                    su += x;
                    sq += x * x;
                }
#pragma unroll
                for (int o = 1; o < CPR; o <<= 1) {
                    su += __shfl_xor(su, o, 64);
                    sq += __shfl_xor(sq, o, 64);
                }
                if (ch == 0 && rg * 32 + row < BMp) {
                    RowP[(colw * BMp + rg * 32 + row) * 2] = su;
                    RowP[(colw * BMp + rg * 32 + row) * 2 + 1] = sq;
                }
            }
            if (ok) {
                const size_t off = (size_t)gm * p.ldc + n0w + ch * 8;
                if constexpr ((flags & FFM_EPI_GELU) != 0) {
                    float a[8];
                    if (p.gelu_deriv) {                          // c = gelu'(x), c2 = gelu(x), x as it would have been stored
                        float d[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) Act<bf16_t>::gelu_both((float)(bf16_t)v[c], a[c], d[c]);
                        Vec8<bf16_t>::store(C + off, d);
                    } else {
                        Vec8<bf16_t>::store(C + off, v);
#pragma unroll
                        for (int c = 0; c < 8; ++c) a[c] = Act<bf16_t>::gelu((float)(bf16_t)v[c]);
                    }
                    Vec8<bf16_t>::store(reinterpret_cast<bf16_t*>(p.c2) + off, a);
                } else {
                    Vec8<bf16_t>::store(C + off, v);
                }
            }
            // (behind the stores: v, the activation chunk and the images' operands are dead here - the tile has 128 registers)
            if constexpr (LNBS) {
                // LNB_STAT: this chunk's share of sum_n dpre[n] pre[n], on the 16-bit values AS STORED (d8s) and the
                // pre-activation chunk in hand: four packed dot products (v_dot2c_f32_bf16 / v_dot2_f32_f16: exact products,
                // fp32 sum) - no conversions, no per-column operands.  The two sums against the fixed vectors W gamma and
                // d = W beta + b are not formed here at all: they ride in the CONSUMER's rank operand as rows 14 / 15
                // (ffm_pack_desc.row14 / row15) and come out of its matrix cores.  The row's CPR chunks meet in the wave's
                // own (dead) LoRA-matrix slab right behind the chunk loop.
                typedef bf16_t h16x2 __attribute__((ext_vector_type(2)));
                float s1 = 0.f, s2 = 0.f;
                {
                    const uint32_t* da = reinterpret_cast<const uint32_t*>(&d8s);
                    const uint32_t* pa = reinterpret_cast<const uint32_t*>(&rpre[rg % PF][i]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const h16x2 x2 = __builtin_bit_cast(h16x2, da[c]), p2 = __builtin_bit_cast(h16x2, pa[c]);
#ifdef FFM_TWIN_F16
                        s2 = __builtin_amdgcn_fdot2(x2, p2, s2, false);
#else
                        s2 = __builtin_amdgcn_fdot2_f32_bf16(x2, p2, s2, false);
#endif
                    }
                }
                f32x2 sq = {s1, s2};
                // chunk (row, ch) has index row * CPR + ch = lane + 64 i in the group's table
                // (the lane id from mbcnt: `lane` itself is not kept live across the epilogue of this 128-register tile)
                const uint32_t ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                const uint32_t qa = (uint32_t)__builtin_amdgcn_readfirstlane((int)lnq) + (ln_ + 64u * i) * 8u;
                asm volatile("ds_write_b64 %0, %1" ::"v"(qa), "v"(sq) : "memory");
            }
        }
        if constexpr (LNBS) {
            // rows of the group: lane r < 32 sums its row's CPR chunks in chunk order (LDS executes a wave's operations in
            // order: the writes above have landed) and leaves the wave's partial in RowP
            const int ln_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            if (ln_ < 32 && rg * 32 + ln_ < BMp) {
                float a1 = 0.f, a2 = 0.f;
                const f32x2* q = reinterpret_cast<const f32x2*>(reinterpret_cast<char*>(LwB) + colw * WN * 64) + ln_ * CPR;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < CPR; ++c) {
                    const f32x2 t = q[c];
                    a1 += t[0];
                    a2 += t[1];
                }
                RowP[(colw * BMp + rg * 32 + ln_) * 2] = a1;
                RowP[(colw * BMp + rg * 32 + ln_) * 2 + 1] = a2;
            }
            asm volatile("" ::: "memory");
        }
        if constexpr (LGRAD) {
            // one operand after the other (registers): V fragment (rank slot frow, rows 8 fgrp .. + 7 of the group) as a
            // bf16 hi + lo pair, the NF column fragments of its image by two transposed reads each, 2 NF MFMAs
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            if constexpr (rg == 0) {
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) {
                    lgc[cf] = lga[cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(lgc[cf]), "+a"(lga[cf]));
                }
            }
            // every LDS read of the phase is issued before the ONE wait (the V tables' packed pairs first, then both images'
            // transposed reads): a wave sits in an LDS round trip once per row group instead of twice
            uint32_t pv[2][8];
#pragma unroll
            for (int ten = 0; ten < 2; ++ten)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    // (rows beyond M are zero in both tables; the phantom rows of an odd MF's last group are clamped AND masked)
                    const int R = rg * 32 + 8 * fgrp + t;
                    const int Rc = R < BMp ? R : BMp - 1;
                    pv[ten][t] = ten ? V2F[Rc * 16 + frow] : reinterpret_cast<const uint32_t*>(V1F)[Rc * r + (frow < r ? frow : 0)];
                }
            panel_u32x2 xr[2][NF][2];
            lg_tr_reads<NF, TROW, 0>(xr[0], t_rd);
            lg_tr_reads<NF, TROW, 1>(xr[1], t_rd);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ten = 0; ten < 2; ++ten) {
                u32x4 ph, pl;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if constexpr (rg == NRG - 1 && (MF & 1)) pv[ten][t] = (rg * 32 + 8 * fgrp + t) < BMp ? pv[ten][t] : 0u;
                    if (!ten) pv[ten][t] = frow < r ? pv[ten][t] : 0u;     // (V2F holds zeros in the slots beyond r)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ph[k] = __builtin_amdgcn_perm(pv[ten][2 * k + 1], pv[ten][2 * k], 0x05040100u);
                    pl[k] = __builtin_amdgcn_perm(pv[ten][2 * k + 1], pv[ten][2 * k], 0x07060302u);
                }
                const frag_t vh = __builtin_bit_cast(frag_t, ph), vl = __builtin_bit_cast(frag_t, pl);
#pragma unroll
                for (int cf = 0; cf < NF; ++cf) {
                    asm volatile("" : "+v"(xr[ten][cf][0]), "+v"(xr[ten][cf][1]));
                    const u32x4 pk = {xr[ten][cf][0][0], xr[ten][cf][0][1], xr[ten][cf][1][0], xr[ten][cf][1][1]};
                    const frag_t xb = __builtin_bit_cast(frag_t, pk);
                    f32x4& dst = ten ? lga[cf] : lgc[cf];
                    // (s_nop 1: an operand the VALU has just written - vh / vl, the zeroed accumulator - needs two wait
                    // states before an MFMA reads it, and hipcc pads nothing for an asm statement: measured, IEEE-half twin)
                    asm volatile("s_nop 1\n\t" FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+a"(dst) : "v"(vh), "v"(xb));
                    asm volatile("s_nop 1\n\t" FFM_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+a"(dst) : "v"(vl), "v"(xb));
                }
            }
        }
        if constexpr (PRE) {
            if constexpr (rg + PF < NRG) load_pre(rg + PF, rpre[rg % PF], rpre2[LNBA ? rg % PF : 0]);
        }
        fence();
    });
    if constexpr (LGRAD) {
        // (asm MFMAs: see behind the main loop.  The accumulators are operands of the wait, or the scheduler lifts their
        // v_accvgpr_read above it, in between the last MFMAs - measured: element 0 of one fragment wrong in full tiles)
        static_assert(NF <= 3 || !LGRAD, "operands of the wait below");
        if constexpr (NF == 3)
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(lgc[0]), "+a"(lgc[1]), "+a"(lgc[2]), "+a"(lga[0]), "+a"(lga[1]), "+a"(lga[2])::"memory");
        else if constexpr (NF == 2)
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(lgc[0]), "+a"(lgc[1]), "+a"(lga[0]), "+a"(lga[1])::"memory");
        else
            asm volatile("s_nop 15\n\ts_nop 15" : "+a"(lgc[0]), "+a"(lga[0])::"memory");
        // D[rank slot 4 fgrp + e][column frow] -> part[tm][n0w + 16 cf + frow][slot]
#pragma unroll
        for (int cf = 0; cf < NF; ++cf) {
            const size_t o = ((size_t)tm * p.N + n0w + cf * 16 + frow) * r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int jj = 4 * fgrp + e;
                if (jj < r) {
                    p.lg_part_c[o + jj] = lgc[cf][e];
                    p.lg_part_a[o + jj] = lga[cf][e];
                }
            }
        }
    }
    if constexpr (LNBS) {
        __syncthreads();
        for (int i = tid; i < BMp; i += ET) {
            if (m0 + i < p.M) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int w = 0; w < CW; ++w) {                        // fixed order over the column slabs
                    a1 += RowP[(w * BMp + i) * 2];
                    a2 += RowP[(w * BMp + i) * 2 + 1];
                }
                f32x2 o = {a1, a2};
                *reinterpret_cast<f32x2*>(p.lnb_part + ((size_t)tn * p.M + m0 + i) * 2) = o;
            }
        }
    }
    if constexpr (ROWST) {
        __syncthreads();
        for (int i = tid; i < BMp; i += ET) {
            if (m0 + i < p.M) {
                float su = 0.f, sq = 0.f;
#pragma unroll
                for (int w = 0; w < CW; ++w) {
                    su += RowP[(w * BMp + i) * 2];
                    sq += RowP[(w * BMp + i) * 2 + 1];
                }
                f32x2 o = {su, sq};
                *reinterpret_cast<f32x2*>(p.rowstat_part + ((size_t)tn * p.M + m0 + i) * 2) = o;
            }
        }
    }
    FFM_STAMP(4);
#ifdef FFM_PANEL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // 5: this wave's output stores have been acknowledged
    FFM_STAMP(5);
#endif
}

template <int MF, int NF, bool RK, int FL, int PWV = 4, int KS = 0>
int launch_panel(const ffm_gemm_args& a, hipStream_t s) {
    constexpr int PW = PWV, PT = PWV * 64, CW = KS ? PW / 2 : PW;
    using G = PanelGeom<MF, RK, PWV>;
    const int tiles = ((a.M + 16 * MF - 1) / (16 * MF)) * (a.N / (CW * 16 * NF));
    constexpr int partb = KS ? CW * (MF * NF + (RK ? (MF + CW - 1) / CW : 0)) * 1024 : 0;      // K split: partial accumulators
    constexpr int lds = (partb > G::RING ? partb : G::RING) +
                        persist_bytes(MF, NF, RK, (FL & (FFM_EPI_LNIN | FFM_EPI_LNB_APPLY)) != 0, CW, (FL & FFM_EPI_LGRAD) != 0,
                                      (FL & FFM_EPI_LNB_STAT) != 0, (FL & FFM_EPI_LNB_APPLY) != 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((RK ? 16 * MF * 192 : 0) + CW * 32 * stage_pitch(NF) * 4 + ((FL & (FFM_EPI_ROWSTATS | FFM_EPI_LNB_STAT)) ? CW * 16 * MF * 8 : 0) <= G::RING,
                  "epilogue tiles alias the ring");
    if (KS && (a.K % 256)) return FFM_EUNSUP;         // the K-split loop is unrolled by four K64 steps
    static bool done = false;                         // one per instantiation
    if (!done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_panel_kernel<MF, NF, RK, FL, PWV, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    hipLaunchKernelGGL((gemm_panel_kernel<MF, NF, RK, FL, PWV, KS>), dim3(tiles), dim3(PT), lds, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

}  // namespace ffm_panel
