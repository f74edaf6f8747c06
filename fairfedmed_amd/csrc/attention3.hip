// Third-generation attention kernels for the vision tower (16-bit storage, no mask, head_dim 64, 65..256 tokens):
// "fat waves" on v_mfma_f32_32x32x16.
//
// What the second generation (attention.hip, attn2_*) spent its time on (profiles/r03_sq_counters.json: MFMA pipes
// 9-11 % busy, half of all wave cycles parked): a wave owned ONE 16-token tile, so every 16 x 32 operand fragment it
// read from LDS fed a single 16x16x32 MFMA, seven 80-register waves per block chased each other through the same
// dependent chain (LDS read -> S -> exp -> pack -> transposed read -> PV), and the whole K / V tile had to land
// before the first MFMA.  Here:
//   * a wave owns a 32-token tile (32 queries in the forward / dQ kernels, 32 keys in the dK/dV kernel) and walks the
//     other index in 32-token tiles with v_mfma_f32_32x32x16: each 1 KiB LDS operand read feeds twice the FLOPs, a
//     score tile is 16 registers per lane, and the per-element VALU work (scale, exp, pack) is issued beside MFMAs
//     that hold the vector issue port for 8 of their 32 cycles instead of 8 of 16;
//   * the first product keeps the OWNED index on the MFMA column (= the lane) and the swept index on the rows
//     (= the registers): S^T = K Q^T in the forward / dQ kernels, S = Q K^T in the dK/dV kernel.  The accumulator
//     is then already the B operand of the second product (cdna_hip_programming.md, "An accumulator tile as the
//     next MFMA's operand"): O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS - nothing crosses LDS
//     or lanes, and row statistics (max, sum, lse, delta) are per-lane scalars in the forward / dQ kernels;
//   * lse and delta enter as the INITIAL accumulators of S and dP (p = exp2(c (s - 8 lse)), dS = p (dP - delta)):
//     one multiply and one exp per element for p, one multiply for dS;
//   * K / V (Q / dO) tiles land row-major in LDS by LDS-DMA in the 8-row x 32-column sub-tile image that serves the
//     ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads without bank conflicts (T10, image (a));
//     the DMA is issued in two halves (first four tiles of both operands, then the rest) behind counted
//     s_waitcnt vmcnt(N) + raw s_barrier, so the first tiles' MFMAs run under the second half's landing;
//   * half a head per block (ceil(NT/2) waves, e.g. 4 + 3 tiles for 197 tokens), block ids b and b + 8 are the two
//     halves of one (batch, head) pair (one XCD under round-robin placement: the second fetch of the tiles is an L2
//     hit - speed only); 768 blocks at 32 images = three per CU, <= 168 registers.
// Rows >= L of a tile are copies of row L - 1 (the DMA source is clamped): finite, and always multiplied by an
// exactly zero probability.  Rows beyond the tile's last 8-row piece are never read: the last tile clamps its rows.
//
// Replaces nn.MultiheadAttention's core (clip/model.py:350-352) and its autograd on the vision tower.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HD = 64;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) char lds_char;

constexpr float A3_C = 0.125f * 1.44269504088896341f;     // log2(e) / sqrt(64)
// Diagnostic builds only (tools/attn_phases.sh): -DFFM_ATTN3_ABL=1 the kernels return once their tiles have landed,
// =2 they issue no tile DMA and compute on whatever LDS holds (results are garbage): prices the two phases.
#ifndef FFM_ATTN3_ABL
#define FFM_ATTN3_ABL 0
#endif
// DMA chunks the forward kernel keeps in flight ahead of the one it computes on.  1: the product build.  2: measured in
// round 5 on a twin library (tools/bench_attn.py, alternating, three pairs): 14.4 / 13.7 / 14.2 us against 14.3 / 13.7 / 13.5 -
// inside the run-to-run spread, not kept
#ifndef FFM_ATTN3_AHEAD
#define FFM_ATTN3_AHEAD 1
#endif
// -DFFM_ATTN3_STAMPS: s_memtime stamps at the phase boundaries of every wave (tools/attn_stamps.py reads them through
// ffm_attn3_read_stamps); the stamps fence the scheduler, so read their SHARES, never this build's run time.
#ifdef FFM_ATTN3_STAMPS
constexpr int A3_NSTAMP = 16;
__device__ unsigned long long a3_stamps[4096 * A3_NSTAMP];
#define A3_STAMP(i)                                                                                      \
    do {                                                                                                 \
        unsigned long long t__;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((threadIdx.x & 63) == 0) a3_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) & 4095) * A3_NSTAMP + (i)] = t__; \
    } while (0)
#define A3_RSTAMP(i)                                                                                     \
    do {                                                                                                 \
        unsigned long long t__;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((threadIdx.x & 63) == 0) a3_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) & 4095) * A3_NSTAMP + (i)] = t__; \
    } while (0)
#else
#define A3_STAMP(i) do { } while (0)
#define A3_RSTAMP(i) do { } while (0)
#endif

template <typename T> struct A3;
template <> struct A3<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct A3<f16_t> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------
// LDS image of a [rows][64] 16-bit tile: 8-row x 32-column sub-tiles of 512 B (cdna_hip_programming.md T10 (a)).
// Byte offset of 16-byte chunk ch (0..7) of row `row`:
__device__ __forceinline__ int img_off(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// One DMA piece = 8 rows x 128 B = 1 KiB, written lane-linearly: lane -> (sub-tile lane >> 5, row (lane >> 2) & 7, slot
// lane & 3); the image's XOR goes onto the SOURCE chunk.  Source rows are clamped to L - 1.
// The LDS-DMA is issued from inline asm (M0 = LDS destination, saved and restored in the same statement:
// cdna_hip_programming.md 5.7): hipcc must not see it, or it drains vmcnt(0) in front of the first
// ds_read_b64_tr_b16 of every phase (measured in the .s), which would serialise the two DMA halves with the compute.
// Every wait for these pieces is a hand-counted s_waitcnt vmcnt(N) followed by a raw s_barrier.
template <typename T>
__device__ __forceinline__ void dma_piece(const T* __restrict__ src, int ld, int L, int pc, char* tile, int lane) {
    if constexpr ((FFM_ATTN3_ABL & 2) != 0) return;
    int row = pc * 8 + ((lane >> 2) & 7);
    const int ch = ((lane >> 5) << 2) + ((lane & 3) ^ ((row >> 2) & 3));
    row = row < L ? row : L - 1;
    const T* g = src + (size_t)row * ld + ch * 8;
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_char*)tile + (uint32_t)pc * 1024u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
}

// A operand of a 32x32x16 MFMA from tile rows (row = 32 t + (lane & 31), k = 16 ks + 8 (lane >> 5) + 0..7):
//   address = tile + 4096 t + 512 (ks >> 1) + (ra ^ (32 (ks & 1))),   ra = img_off(lane & 31, lane >> 5)
__device__ __forceinline__ int row_lane_off(int lane) { return img_off(lane & 31, lane >> 5); }
template <typename T>
__device__ __forceinline__ typename A3<T>::frag row_frag(const char* tile, int ra, int t, int ks) {
    return *reinterpret_cast<const typename A3<T>::frag*>(tile + ((ra ^ (32 * (ks & 1))) + 4096 * t + 512 * (ks >> 1)));
}
template <typename T>
__device__ __forceinline__ typename A3<T>::frag row_frag_clamped(const char* tile, int lane, int t, int ks, int rmax) {
    int row = 32 * t + (lane & 31);
    row = row < rmax ? row : rmax;
    return *reinterpret_cast<const typename A3<T>::frag*>(tile + img_off(row, 2 * ks + (lane >> 5)));
}

// A operand X^T [32 columns of X (d = 32 dt + (lane & 31))][16 rows of X] for the k-step s of token tile t, with the k
// order of an accumulator used as the B operand: element j of lane half h is token 32 t + 16 s + 8 (j >> 2) + 4 h + (j & 3).
// Two transposed reads (T10): lane 4q + p of a 16-lane group G addresses row q, columns 4p .. 4p + 3 of the group's
// 4 x 16 block (rows 32 t + 16 s + 4 (G >> 1) [+ 8], columns 32 dt + 16 (G & 1)):
//   address = tile + 4096 t + 2048 s + 512 dt + ta            (tokens + 0..3)
//           = tile + 4096 t + 2048 s + 512 dt + 1024 + (ta ^ 32)   (tokens + 8..11: next 8-row group, swizzle ^ 2)
__device__ __forceinline__ int tr_lane_off(int lane) {
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = G >> 1;
    return 64 * (4 * h + q) + 16 * ((2 * (G & 1) + (p >> 1)) ^ h) + 8 * (p & 1);
}
template <typename T>
__device__ __forceinline__ typename A3<T>::frag tr_frag(const char* tile, int ta, int t, int s, int dt) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const int o = 4096 * t + 2048 * s + 512 * dt;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(tile + (ta + o)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(tile + ((ta ^ 32) + o + 1024)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename A3<T>::frag, v);
}
template <typename T>
__device__ __forceinline__ typename A3<T>::frag tr_frag_clamped(const char* tile, int lane, int t, int s, int dt, int rmax) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const int G = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = G >> 1;
    int r0 = 32 * t + 16 * s + 4 * h + q, r1 = r0 + 8;
    r0 = r0 < rmax ? r0 : rmax;
    r1 = r1 < rmax ? r1 : rmax;
    const int ch = 4 * dt + 2 * (G & 1) + (p >> 1), b = 8 * (p & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(tile + img_off(r0, ch) + b));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(tile + img_off(r1, ch) + b));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename A3<T>::frag, v);
}

// registers 8 s .. 8 s + 7 of a 32 x 32 accumulator -> the B operand of k-step s
template <typename T>
__device__ __forceinline__ typename A3<T>::frag pack8(const f32x16& a, int s) {
    typename A3<T>::frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (T)a[8 * s + j];
    return f;
}

// the other 32-lane half's value (rows of a 32 x 32 accumulator are split 4 h + ... over the two halves)
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// token row of register r of a 32 x 32 accumulator for lane half h
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }   // + 4 h

// (batch, head) pair and half of a block id.  Block ids b, b + 8, b + 16, ... share an XCD under the round-robin placement
// observed (speed only: any placement computes the same), so XCD label x = b & 7 takes the CONTIGUOUS range of units
// [x U/8, (x + 1) U/8): the two halves of a pair sit on one XCD (the second fetch of the tiles is an L2 hit), and so do all
// heads of an image - the images whose qkv rows (and dO rows in the backward) the panel GEMM in front of this kernel produced
// on the same XCD (its xcd_remap gives XCD x whole row bands: ~4 images of 197 rows at batch 32).  FFM_ATTN3_MAP=0 (A/B
// runs) spreads an image's heads over the XCDs as rounds 2-3 did.  Which unit of a pair is the larger half flips every 32
// units of an XCD: a CU hosts units j, j + 32, j + 64 of its XCD's range, and three larger halves (12 tiles) beside three
// smaller ones (9) on the next CU cost the launch its balance.
__device__ __forceinline__ void unit_of_block(int bid, int& bh, int& part, int map) {
    if (map) {
        const int per = gridDim.x >> 3, j = bid >> 3;
        const int u = (bid & 7) * per + j;
        bh = u >> 1;
        part = (u ^ (j >> 5)) & 1;
    } else {
        bh = (bid >> 4) * 8 + (bid & 7);
        part = ((bid >> 3) ^ (bid >> 8)) & 1;
    }
}

// 16-byte operand fragments straight from global memory, hidden from hipcc's s_waitcnt bookkeeping: beside LDS-DMA in
// flight it would wait vmcnt(0) at their first use (cdna_hip_programming.md section 5, "Pipelining across barriers").
// The caller waits with wait_frags<N>() (which names every destination) before the first use.
template <int OFF>
__device__ __forceinline__ void asm_load16(u32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void asm_load4(float& dst, const void* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_frags(u32x4 (&a)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ void block_sync() {
    // raw barrier: __syncthreads() would drain the LDS-DMA still in flight (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// a 32-row x 64-column tile of the (normalised / scaled) transposed accumulators -> rows of a token-major matrix.
// Register group i of d-tile dt holds columns 32 dt + 8 i + 4 h + 0..3 of row (lane & 31): pairs of groups are swapped
// between the lane halves (v_permlane32_swap, T21) so that every lane stores 16 contiguous bytes.
template <typename T>
__device__ __forceinline__ void store_rows(T* __restrict__ row_ptr, const f32x16 (&o)[2], float scale, int lane) {
    const int hb = (lane >> 5) * 16;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 a = {(T)(o[dt][4 * i] * scale), (T)(o[dt][4 * i + 1] * scale), (T)(o[dt][4 * i + 2] * scale), (T)(o[dt][4 * i + 3] * scale)};
            t4 b = {(T)(o[dt][4 * i + 4] * scale), (T)(o[dt][4 * i + 5] * scale), (T)(o[dt][4 * i + 6] * scale), (T)(o[dt][4 * i + 7] * scale)};
            u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
            const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(row_ptr) + 64 * dt + 16 * i + hb) = v;
        }
}

// The same store, also forming this lane's share of the two LayerNorm-backward row sums of ln_1 (include/ffm_hip.h,
// FFM_EPI_LNB_*; the qkv product is LayerNorm-folded, so with dqkv the gradient of its output and x the stored qkv values
//   c1 = sum_n dqkv[n] (W gamma)[n],   c2 = sum_n dqkv[n] (qkv[n] - d[n]),   d = W beta + b ):
// the 16-byte chunk a lane stores (8 consecutive head dims 16 ks + 8 h ..) lines up with the fragment xf[ks] of the same row
// it has held since the prologue; tab = {W gamma, d} of this 64-column slice in LDS ([2][64] floats).  s1 / s2 accumulate over
// the lane's 32 head dims; the other 32 are in lane ^ 32.
template <typename T>
__device__ __forceinline__ void store_rows_stat(T* __restrict__ row_ptr, const f32x16 (&o)[2], float scale, int lane,
                                                const typename A3<T>::frag (&xf)[4], const float* tab, float& s1, float& s2) {
    const int hb = (lane >> 5) * 16;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            typedef __attribute__((ext_vector_type(4))) T t4;
            typedef __attribute__((ext_vector_type(8))) T t8;
            t4 a = {(T)(o[dt][4 * i] * scale), (T)(o[dt][4 * i + 1] * scale), (T)(o[dt][4 * i + 2] * scale), (T)(o[dt][4 * i + 3] * scale)};
            t4 b = {(T)(o[dt][4 * i + 4] * scale), (T)(o[dt][4 * i + 5] * scale), (T)(o[dt][4 * i + 6] * scale), (T)(o[dt][4 * i + 7] * scale)};
            u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
            const auto r0 = __builtin_amdgcn_permlane32_swap(ua[0], ub[0], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(ua[1], ub[1], false, false);
            const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(row_ptr) + 64 * dt + 16 * i + hb) = v;
            const t8 g8 = __builtin_bit_cast(t8, v);                    // the 8 values AS STORED: head dims 32 dt + 8 i + 8 h + 0..7
            const int ks = 2 * dt + i / 2, d0 = 16 * ks + (hb >> 1);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(tab + d0), w1 = *reinterpret_cast<const f32x4*>(tab + d0 + 4);
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(tab + 64 + d0), e1 = *reinterpret_cast<const f32x4*>(tab + 64 + d0 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = (float)g8[j];
                s1 += g * (j < 4 ? w0[j & 3] : w1[j & 3]);
                s2 += g * ((float)xf[ks][j] - (j < 4 ? e0[j & 3] : e1[j & 3]));
            }
        }
}

template <int NT> struct Geo {
    static constexpr int NW = (NT + 1) / 2;                     // waves per block = 32-token tiles of the larger half
    static constexpr int NCH = (NT + 1) / 2;                    // DMA chunks of two swept tiles (64 rows = 8 pieces per operand)
    static constexpr int RC = (8 + NW - 1) / NW;                // piece rounds per chunk and operand (waves that run out repeat a piece)
    static constexpr int CA = NT < 4 ? NT : 4, CB = NT - CA;    // swept tiles of the forward's two softmax halves
};

// Pieces of DMA chunk c (rows 64 c .. 64 c + 63 of the two operand tiles), dealt round-robin to the NW waves.  The chunks are
// issued ONE AHEAD of the compute ([issue c + 1][wait c][barrier][compute c]): with every piece of a block issued up front,
// the waves sat in the ISSUE of their DMA for the whole load (in-kernel stamps: 6 400 of a wave's 17 700 cycles; the CU's
// memory path holds ~70 KB in flight, three blocks ask for 153 KB) and the first MFMA waited for the last byte.
template <typename T, int NT>
__device__ __forceinline__ void dma_chunk(int c, const T* __restrict__ s0, int ld0, char* t0, const T* __restrict__ s1, int ld1, char* t1,
                                          int L, int NP, int wave, int lane) {
    typedef Geo<NT> GE;
#pragma unroll
    for (int i = 0; i < GE::RC; ++i) {
        int pc = 8 * c + wave + GE::NW * i;
        pc = pc < 8 * c + 7 ? pc : 8 * c + 7;
        pc = pc < NP ? pc : NP - 1;
        dma_piece<T>(s0, ld0, L, pc, t0, lane);
        dma_piece<T>(s1, ld1, L, pc, t1, lane);
    }
}

// ---------------------------------------------------------------------------
// forward: a wave owns 32 queries; S^T = K Q^T per 32-key tile, running maximum over two halves of the keys, O^T = V^T P^T
// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(64 * Geo<NT>::NW) __attribute__((amdgpu_waves_per_eu(3, 3)))
void attn3_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse, int L, int heads, int BH, int map) {
    typedef typename A3<T>::frag frag;
    typedef Geo<NT> GE;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int R8 = (L + 7) & ~7, NP = R8 >> 3, rmax = R8 - 1;
    char* Ks = smem;                                            // [R8][128 B]; reads past it land in Vs
    char* Vs = smem + R8 * 128;
    int bh, part;
    unit_of_block(blockIdx.x, bh, part, map);
    if (bh >= BH) return;
    const int b = bh / heads, hd = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + hd * HD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n0 = (NT + 1) / 2;
    const int qt = part ? n0 + wave : wave;
    const bool active = part ? wave < NT - n0 : wave < n0;
    const int q = qt * 32 + r;

    A3_RSTAMP(14);
    A3_STAMP(0);
    u32x4 qv[4];
    {
        const int qr = active && q < L ? q : L - 1;
        const T* qp = base + (size_t)qr * ld + 8 * h;
        asm_load16<0>(qv[0], qp);
        asm_load16<32>(qv[1], qp);
        asm_load16<64>(qv[2], qp);
        asm_load16<96>(qv[3], qp);
    }
    dma_chunk<T, NT>(0, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
    if constexpr ((FFM_ATTN3_ABL & 1) != 0) {
#pragma unroll
        for (int c = 1; c < GE::NCH; ++c) dma_chunk<T, NT>(c, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
        wait_frags<0>(qv);
        return;
    }
    if constexpr (FFM_ATTN3_AHEAD > 1 && GE::NCH > 1) dma_chunk<T, NT>(1, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
    A3_STAMP(1);

    const int ra = row_lane_off(lane), ta = tr_lane_off(lane);
    const bool skip_last_step = L - 32 * (NT - 1) <= 16;        // the last tile's second k-step holds no key < L
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    frag qf[4];

    // chunk c has landed in every wave's view: issue the next one first (its issue stalls on the memory path about as long
    // as this wait would anyway), then the counted wait for c, then the barrier
    auto chunk_ready = [&](auto C_) {
        constexpr int c = decltype(C_)::value;
        constexpr int AH = FFM_ATTN3_AHEAD;
        if constexpr (c + AH < GE::NCH) dma_chunk<T, NT>(c + AH, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
        constexpr int last = c + AH < GE::NCH - 1 ? c + AH : GE::NCH - 1;      // youngest chunk in flight
        constexpr int pend = 2 * GE::RC * (last - c);
        if constexpr (c == 0) {
            wait_frags<pend>(qv);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(frag, qv[ks]);
        } else {
            wait_vm<pend>();
        }
        block_sync();
    };

    auto half = [&](auto F0_, auto CF_, auto FIRST_) {
        constexpr int F0 = decltype(F0_)::value, CF = decltype(CF_)::value;
        constexpr bool FIRST = decltype(FIRST_)::value;
        f32x16 s[CF];
        float mx = m;
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            if ((F0 + f) % 2 == 0) {                            // tile F0 + f opens DMA chunk (F0 + f) / 2
                if ((F0 + f) / 2 == 0) chunk_ready(std::integral_constant<int, 0>{});
                if ((F0 + f) / 2 == 1) chunk_ready(std::integral_constant<int, (GE::NCH > 1 ? 1 : 0)>{});
                if ((F0 + f) / 2 == 2) chunk_ready(std::integral_constant<int, (GE::NCH > 2 ? 2 : 0)>{});
                if ((F0 + f) / 2 == 3) chunk_ready(std::integral_constant<int, (GE::NCH > 3 ? 3 : 0)>{});
                if (F0 + f == 0) A3_STAMP(2);
            }
            if (!active) continue;
            const int t = F0 + f;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = A3<T>::mma(row_frag<T>(Ks, ra, t, ks), qf[ks], acc);
            if (F0 + f == NT - 1) {                             // only the last key tile can straddle L
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (32 * (NT - 1) + acc_row(e) + 4 * h >= L) acc[e] = -INFINITY;
            }
            s[f] = acc;
#pragma unroll
            for (int e = 0; e < 16; e += 2) mx = fmaxf(mx, fmaxf(acc[e], acc[e + 1]));
        }
        if (!active) return;
        mx = fmaxf(mx, xhalf(mx));
        const float mc = mx * A3_C;
        if constexpr (!FIRST) {
            // running maximum: everything accumulated so far is rescaled by 2^(c (m - mx))
            const float alpha = __builtin_amdgcn_exp2f((m - mx) * A3_C);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
        m = mx;
        float l4[4] = {0.f, 0.f, 0.f, 0.f};                     // four partial sums: no 64-deep chain of dependent adds
#pragma unroll
        for (int f = 0; f < CF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[f][e], A3_C, -mc));
                s[f][e] = p;
                l4[e & 3] += p;
            }
        l += (l4[0] + l4[1]) + (l4[2] + l4[3]);
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            const int t = F0 + f;
            if (F0 + f == NT - 1) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    if (st == 1 && skip_last_step) break;
                    const frag pf = pack8<T>(s[f], st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = A3<T>::mma(tr_frag_clamped<T>(Vs, lane, t, st, dt, rmax), pf, o[dt]);
                }
            } else {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const frag pf = pack8<T>(s[f], st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) o[dt] = A3<T>::mma(tr_frag<T>(Vs, ta, t, st, dt), pf, o[dt]);
                }
            }
        }
    };
    half(std::integral_constant<int, 0>{}, std::integral_constant<int, GE::CA>{}, std::true_type{});
    if constexpr (GE::CB > 0) half(std::integral_constant<int, GE::CA>{}, std::integral_constant<int, GE::CB>{}, std::false_type{});
    A3_STAMP(3);
    if (!active) return;

    l += xhalf(l);
    if (q < L) {
        store_rows<T>(out + ((size_t)b * L + q) * E + hd * HD, o, 1.0f / l, lane);
        if (h == 0 && lse) lse[((size_t)b * heads + hd) * L + q] = m * 0.125f + __logf(l);
    }
    A3_STAMP(4);
    A3_RSTAMP(15);
}

// ---------------------------------------------------------------------------
// dQ (and delta = rowsum(dO * O), which the dK/dV kernel reads back): a wave owns 32 queries.
// S^T = K Q^T - 8 lse and dP^T = V dO^T - delta per 32-key tile, dS^T = exp2(c S^T) dP^T, dQ^T += K^T dS^T.
// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(64 * Geo<NT>::NW) __attribute__((amdgpu_waves_per_eu(3, 3)))
void attn3_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o, const float* __restrict__ lse,
                         const T* __restrict__ o_fwd, T* __restrict__ dqkv, float* __restrict__ delta, int L, int heads, int BH, int map,
                         const float* __restrict__ ln_wg, const float* __restrict__ ln_d, float* __restrict__ ln_part) {
    typedef typename A3<T>::frag frag;
    typedef Geo<NT> GE;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int R8 = (L + 7) & ~7, NP = R8 >> 3, rmax = R8 - 1;
    char* Ks = smem;
    char* Vs = smem + R8 * 128;
    float* tabs = reinterpret_cast<float*>(smem + 2 * R8 * 128);                // ln_part: {W gamma, d} of this head's q columns, [2][64]
    int bh, part;
    unit_of_block(blockIdx.x, bh, part, map);
    if (bh >= BH) return;
    const int b = bh / heads, hd = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + hd * HD;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n0 = (NT + 1) / 2;
    const int qt = part ? n0 + wave : wave;
    const bool active = part ? wave < NT - n0 : wave < n0;
    const int q = qt * 32 + r;
    const int qr = active && q < L ? q : L - 1;

    u32x4 qv[4], dov[4], ov[4];
    float lq;
    {
        const T* qp = base + (size_t)qr * ld + 8 * h;
        const T* dp = d_o + ((size_t)b * L + qr) * E + hd * HD + 8 * h;
        const T* op = o_fwd + ((size_t)b * L + qr) * E + hd * HD + 8 * h;
        asm_load16<0>(qv[0], qp); asm_load16<32>(qv[1], qp); asm_load16<64>(qv[2], qp); asm_load16<96>(qv[3], qp);
        asm_load16<0>(dov[0], dp); asm_load16<32>(dov[1], dp); asm_load16<64>(dov[2], dp); asm_load16<96>(dov[3], dp);
        asm_load16<0>(ov[0], op); asm_load16<32>(ov[1], op); asm_load16<64>(ov[2], op); asm_load16<96>(ov[3], op);
        asm_load4(lq, lse + ((size_t)b * heads + hd) * L + qr);
    }
    // ln_1's backward row sums (ln_part != NULL, uniform): threads 0..31 fetch the two 64-float tables of this head's q columns
    u32x4 tabv = {0u, 0u, 0u, 0u};
    if (ln_part) {
        const int ti = threadIdx.x & 31;
        asm_load16<0>(tabv, (ti < 16 ? ln_wg : ln_d) + hd * HD + 4 * (ti & 15));
    }
    dma_chunk<T, NT>(0, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
    constexpr bool LOADS_ONLY = (FFM_ATTN3_ABL & 1) != 0;
    if constexpr (LOADS_ONLY || GE::NCH > 1) {
#pragma unroll
        for (int c = 1; c < (LOADS_ONLY ? GE::NCH : 2); ++c) dma_chunk<T, NT>(c, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
    }
    const int ra = row_lane_off(lane), ta = tr_lane_off(lane);
    const bool skip_last_step = L - 32 * (NT - 1) <= 16;

    // the 13 fragment loads are older than every DMA piece: they are back when chunk 0's pieces are
    asm volatile("s_waitcnt vmcnt(%13)"
                 : "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3]), "+v"(dov[0]), "+v"(dov[1]), "+v"(dov[2]), "+v"(dov[3]),
                   "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(lq)
                 : "n"(LOADS_ONLY ? 0 : GE::NCH > 1 ? 2 * GE::RC : 0)
                 : "memory");
    if constexpr (LOADS_ONLY) return;
    // (the table load was issued behind the fragment loads and IN FRONT of the DMA pieces of chunks 0 / 1: the counted wait
    // above has it too - vmcnt retires in order)
    asm volatile("" : "+v"(tabv));
    if (ln_part && threadIdx.x < 32) reinterpret_cast<u32x4*>(tabs)[threadIdx.x] = tabv;      // visible behind chunk 0's barrier
    frag qf[4], dof[4];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = __builtin_bit_cast(frag, qv[ks]);
        dof[ks] = __builtin_bit_cast(frag, dov[ks]);
        const frag of = __builtin_bit_cast(frag, ov[ks]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)of[e] * (float)dof[ks][e];
    }
    dl += xhalf(dl);
    if (active && h == 0 && q < L) delta[((size_t)b * heads + hd) * L + q] = dl;
    const float s0 = -8.0f * lq, p0 = -dl;

    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

    auto tile = [&](auto T_) {
        constexpr int t = decltype(T_)::value;
        constexpr bool LAST = t == NT - 1;
        f32x16 sa, pa;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sa[e] = s0; pa[e] = p0; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            sa = A3<T>::mma(row_frag<T>(Ks, ra, t, ks), qf[ks], sa);           // (the last tile reads on into Vs: masked)
            if constexpr (LAST) pa = A3<T>::mma(row_frag_clamped<T>(Vs, lane, t, ks, rmax), dof[ks], pa);
            else pa = A3<T>::mma(row_frag<T>(Vs, ra, t, ks), dof[ks], pa);
        }
        if constexpr (LAST) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (32 * (NT - 1) + acc_row(e) + 4 * h >= L) sa[e] = -INFINITY;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sa[e] = __builtin_amdgcn_exp2f(sa[e] * A3_C) * pa[e];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (LAST && st == 1 && skip_last_step) break;
            const frag df = pack8<T>(sa, st);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if constexpr (LAST) dq[dt] = A3<T>::mma(tr_frag_clamped<T>(Ks, lane, t, st, dt, rmax), df, dq[dt]);
                else dq[dt] = A3<T>::mma(tr_frag<T>(Ks, ta, t, st, dt), df, dq[dt]);
            }
        }
    };
    // chunk c: [issue c + 2 ... no: c + 1 was issued one step ago][wait c][barrier][tiles 2c, 2c + 1]
    auto chunk = [&](auto C_) {
        constexpr int c = decltype(C_)::value;
        if constexpr (c > 0) {                                                  // (chunk 1 rode with the prologue)
            if constexpr (c + 1 < GE::NCH) dma_chunk<T, NT>(c + 1, base + E, ld, Ks, base + 2 * E, ld, Vs, L, NP, wave, lane);
            wait_vm<(c + 1 < GE::NCH ? 2 * GE::RC : 0)>();
        }
        block_sync();
        if (active) {
            tile(std::integral_constant<int, 2 * c>{});
            if constexpr (2 * c + 1 < NT) tile(std::integral_constant<int, 2 * c + 1>{});
        }
    };
    chunk(std::integral_constant<int, 0>{});
    if constexpr (GE::NCH > 1) chunk(std::integral_constant<int, 1>{});
    if constexpr (GE::NCH > 2) chunk(std::integral_constant<int, 2>{});
    if constexpr (GE::NCH > 3) chunk(std::integral_constant<int, 3>{});
    if (ln_part) {
        float s1 = 0.f, s2 = 0.f;
        if (active && q < L) store_rows_stat<T>(dqkv + ((size_t)b * L + q) * ld + hd * HD, dq, 0.125f, lane, qf, tabs, s1, s2);
        s1 += xhalf(s1);
        s2 += xhalf(s2);
        // partial row [slot = head] of the q columns; the k and v columns' [heads + head] comes from the dK/dV kernel
        if (active && q < L && h == 0) {
            f32x2 o2 = {s1, s2};
            *reinterpret_cast<f32x2*>(ln_part + ((size_t)hd * ((size_t)(BH / heads) * L) + (size_t)b * L + q) * 2) = o2;
        }
    } else if (active && q < L) {
        store_rows<T>(dqkv + ((size_t)b * L + q) * ld + hd * HD, dq, 0.125f, lane);
    }
}

// ---------------------------------------------------------------------------
// dK / dV: a wave owns 32 keys and sweeps the queries in 32-row tiles, the key on the MFMA column (= lane):
// S = Q K^T - 8 lse[q], dP = dO V^T - delta[q] (row constants as the initial accumulators, from LDS), P = exp2(c S),
// dS = P dP; dV^T += dO^T P and dK^T += Q^T dS take P / dS straight from the accumulators and the transposed Q / dO
// operands from the same row-major tiles.
// ---------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(64 * Geo<NT>::NW) __attribute__((amdgpu_waves_per_eu(3, 3)))
void attn3_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o, const float* __restrict__ lse,
                          const float* __restrict__ delta, T* __restrict__ dqkv, int L, int heads, int BH, int map,
                          const float* __restrict__ ln_wg, const float* __restrict__ ln_d, float* ln_part) {
    typedef typename A3<T>::frag frag;
    typedef Geo<NT> GE;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int R8 = (L + 7) & ~7, NP = R8 >> 3, rmax = R8 - 1;
    char* Qs = smem;
    char* dOs = smem + R8 * 128;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * R8 * 128);              // [32 NT]: -8 lse (-inf beyond L)
    float* del_s = lse_s + 32 * NT;                                             // [32 NT]: -delta (0 beyond L)
    // ln_part: {W gamma, d} of the k | v columns, [2][2][64] floats = 1 KB - laid over lse_s / del_s once the last tile is done
    // (32 NT >= 128 floats each): a block of its own would take the kernel from three blocks per CU to two (LDS is dealt in
    // 1280-byte units: 52 992 B -> 42 units, 54 016 B -> 43), which cost 4.8 us per launch when it was tried
    float* tabs = lse_s;
    constexpr bool LN_OK = 2 * 32 * NT >= 256;                                   // (NT = 3: not served - ffm_attention_bwd_lnstat_ok)
    if (!LN_OK) ln_part = nullptr;
    int bh, part;
    unit_of_block(blockIdx.x, bh, part, map);
    if (bh >= BH) return;
    const int b = bh / heads, hd = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + hd * HD;
    const T* dob = d_o + (size_t)b * L * E + hd * HD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int n0 = (NT + 1) / 2;
    const int kt = part ? n0 + wave : wave;
    const bool active = part ? wave < NT - n0 : wave < n0;
    const int key = kt * 32 + r;

    A3_RSTAMP(14);
    A3_STAMP(0);
    u32x4 kv[4], vv[4];
    float lv[2], dv_[2];
    constexpr int NTH = 64 * GE::NW, NLD = (32 * NT + NTH - 1) / NTH;           // lse / delta values per thread (<= 2)
    static_assert(NLD <= 2, "row constants: at most two per thread");
    {
        const int kr = active && key < L ? key : L - 1;
        const T* kp = base + E + (size_t)kr * ld + 8 * h;
        const T* vp = base + 2 * E + (size_t)kr * ld + 8 * h;
        asm_load16<0>(kv[0], kp); asm_load16<32>(kv[1], kp); asm_load16<64>(kv[2], kp); asm_load16<96>(kv[3], kp);
        asm_load16<0>(vv[0], vp); asm_load16<32>(vv[1], vp); asm_load16<64>(vv[2], vp); asm_load16<96>(vv[3], vp);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * NTH;
            const size_t o = ((size_t)b * heads + hd) * L + (i < L ? i : L - 1);
            asm_load4(lv[j], lse + o);
            asm_load4(dv_[j], delta + o);
        }
    }
    u32x4 tabv = {0u, 0u, 0u, 0u};
    if (ln_part) {                                                              // threads 0..63: k tables, then v tables
        const int ti = tid & 63, part_ = ti >> 5, tj = ti & 31;
        asm_load16<0>(tabv, (tj < 16 ? ln_wg : ln_d) + (1 + part_) * E + hd * HD + 4 * (tj & 15));
    }
    dma_chunk<T, NT>(0, base, ld, Qs, dob, E, dOs, L, NP, wave, lane);
    constexpr bool LOADS_ONLY = (FFM_ATTN3_ABL & 1) != 0;
    if constexpr (LOADS_ONLY || GE::NCH > 1) {
#pragma unroll
        for (int c = 1; c < (LOADS_ONLY ? GE::NCH : 2); ++c) dma_chunk<T, NT>(c, base, ld, Qs, dob, E, dOs, L, NP, wave, lane);
    }
    const int ra = row_lane_off(lane), ta = tr_lane_off(lane);
    const bool skip_last_step = L - 32 * (NT - 1) <= 16;

    asm volatile("s_waitcnt vmcnt(%12)"
                 : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]), "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]),
                   "+v"(lv[0]), "+v"(lv[1]), "+v"(dv_[0]), "+v"(dv_[1])
                 : "n"(LOADS_ONLY ? 0 : GE::NCH > 1 ? 2 * GE::RC : 0)
                 : "memory");
    if constexpr (LOADS_ONLY) return;
    asm volatile("" : "+v"(tabv));                                               // (kept in 4 registers until the tiles are done)
    A3_STAMP(1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = tid + j * NTH;
        if (i < 32 * NT) {
            lse_s[i] = i < L ? -8.0f * lv[j] : -INFINITY;
            del_s[i] = i < L ? -dv_[j] : 0.f;
        }
    }
    frag kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = __builtin_bit_cast(frag, kv[ks]);
        vf[ks] = __builtin_bit_cast(frag, vv[ks]);
    }
    f32x16 dv[2], dk[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dv[dt][e] = 0.f; dk[dt][e] = 0.f; }

    auto tile = [&](auto T_) {
        constexpr int t = decltype(T_)::value;
        constexpr bool LAST = t == NT - 1;
        f32x16 sa, pa;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(&lse_s[32 * t + 8 * i + 4 * h]);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(&del_s[32 * t + 8 * i + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sa[4 * i + e] = l4[e]; pa[4 * i + e] = d4[e]; }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (LAST) {
                sa = A3<T>::mma(row_frag_clamped<T>(Qs, lane, t, ks, rmax), kf[ks], sa);
                pa = A3<T>::mma(row_frag_clamped<T>(dOs, lane, t, ks, rmax), vf[ks], pa);
            } else {
                sa = A3<T>::mma(row_frag<T>(Qs, ra, t, ks), kf[ks], sa);
                pa = A3<T>::mma(row_frag<T>(dOs, ra, t, ks), vf[ks], pa);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sa[e] = __builtin_amdgcn_exp2f(sa[e] * A3_C);
            pa[e] *= sa[e];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (LAST && st == 1 && skip_last_step) break;
            const frag pf = pack8<T>(sa, st), df = pack8<T>(pa, st);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if constexpr (LAST) {
                    dv[dt] = A3<T>::mma(tr_frag_clamped<T>(dOs, lane, t, st, dt, rmax), pf, dv[dt]);
                    dk[dt] = A3<T>::mma(tr_frag_clamped<T>(Qs, lane, t, st, dt, rmax), df, dk[dt]);
                } else {
                    dv[dt] = A3<T>::mma(tr_frag<T>(dOs, ta, t, st, dt), pf, dv[dt]);
                    dk[dt] = A3<T>::mma(tr_frag<T>(Qs, ta, t, st, dt), df, dk[dt]);
                }
            }
        }
    };
    auto chunk = [&](auto C_) {
        constexpr int c = decltype(C_)::value;
        if constexpr (c > 0) {                                                  // (chunk 1 rode with the prologue)
            if constexpr (c + 1 < GE::NCH) dma_chunk<T, NT>(c + 1, base, ld, Qs, dob, E, dOs, L, NP, wave, lane);
            wait_vm<(c + 1 < GE::NCH ? 2 * GE::RC : 0)>();
        }
        block_sync();                                                           // (chunk 0: also lse_s, del_s)
        if constexpr (c == 0) A3_STAMP(2);
        if (active) {
            tile(std::integral_constant<int, 2 * c>{});
            if constexpr (2 * c + 1 < NT) tile(std::integral_constant<int, 2 * c + 1>{});
        }
    };
    chunk(std::integral_constant<int, 0>{});
    if constexpr (GE::NCH > 1) chunk(std::integral_constant<int, 1>{});
    if constexpr (GE::NCH > 2) chunk(std::integral_constant<int, 2>{});
    if constexpr (GE::NCH > 3) chunk(std::integral_constant<int, 3>{});
    A3_STAMP(3);
    if (ln_part) {
        block_sync();                                                           // every wave is done with lse_s / del_s
        asm volatile("" : "+v"(tabv));
        if (tid < 64) reinterpret_cast<u32x4*>(tabs)[tid] = tabv;               // [k: wg, d][v: wg, d]
        block_sync();
        float s1 = 0.f, s2 = 0.f;
        if (active && key < L) {
            T* drow = dqkv + ((size_t)b * L + key) * ld + hd * HD;
            store_rows_stat<T>(drow + E, dk, 0.125f, lane, kf, tabs, s1, s2);
            store_rows_stat<T>(drow + 2 * E, dv, 1.0f, lane, vf, tabs + 128, s1, s2);
        }
        s1 += xhalf(s1);
        s2 += xhalf(s2);
        if (active && key < L && h == 0) {
            f32x2 o2 = {s1, s2};
            *reinterpret_cast<f32x2*>(ln_part + ((size_t)(heads + hd) * ((size_t)(BH / heads) * L) + (size_t)b * L + key) * 2) = o2;
        }
    } else if (active && key < L) {
        T* drow = dqkv + ((size_t)b * L + key) * ld + hd * HD;
        store_rows<T>(drow + E, dk, 0.125f, lane);
        store_rows<T>(drow + 2 * E, dv, 1.0f, lane);
    }
    A3_STAMP(4);
    A3_RSTAMP(15);
}

inline int a3_map() {
    static const int m = (getenv("FFM_ATTN3_MAP") && getenv("FFM_ATTN3_MAP")[0] == '0') ? 0 : 1;
    return m;
}

template <typename F> int set_lds3(F fn, int bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <typename T, int NT>
int run_fwd3(const void* qkv, void* out, float* lse, int B, int L, int heads, hipStream_t s) {
    const int R8 = (L + 7) & ~7, lds = 2 * R8 * 128, BH = B * heads;
    int e = set_lds3(attn3_fwd_kernel<T, NT>, lds);
    if (e) return e;
    hipLaunchKernelGGL((attn3_fwd_kernel<T, NT>), dim3(((BH + 7) / 8) * 16), dim3(64 * Geo<NT>::NW), lds, s, (const T*)qkv, (T*)out, lse, L,
                       heads, BH, a3_map());
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
template <typename T, int NT>
int run_bwd3(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L, int heads,
             hipStream_t s, const float* ln_wg, const float* ln_d, float* ln_part) {
    // (+ 512 B in the dQ kernel: the LayerNorm-backward tables of the head - 51 712 B is still 41 of the 1280-byte LDS units,
    // three blocks per CU; the dK/dV kernel lays its 1 KB over the row constants at its end, see there)
    const int R8 = (L + 7) & ~7, lds_dq = 2 * R8 * 128 + 512, lds_dkv = 2 * R8 * 128 + 2 * 32 * NT * 4, BH = B * heads;
    int e = set_lds3(attn3_bwd_dq_kernel<T, NT>, lds_dq);
    if (e) return e;
    e = set_lds3(attn3_bwd_dkv_kernel<T, NT>, lds_dkv);
    if (e) return e;
    const dim3 grid(((BH + 7) / 8) * 16), block(64 * Geo<NT>::NW);
    // (the dK/dV kernel on a second stream BESIDE the dQ kernel - a timing probe with a stale delta, to price ONE launch
    // holding both kernels' blocks with delta recomputed in the dK/dV blocks: 4.74 against 4.72 ms per step, three alternating
    // pairs - nothing)
    hipLaunchKernelGGL((attn3_bwd_dq_kernel<T, NT>), grid, block, lds_dq, s, (const T*)qkv, (const T*)dout, lse, (const T*)out, (T*)dqkv,
                       delta, L, heads, BH, a3_map(), ln_wg, ln_d, ln_part);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn3_bwd_dkv_kernel<T, NT>), grid, block, lds_dkv, s, (const T*)qkv, (const T*)dout, lse, (const float*)delta,
                       (T*)dqkv, L, heads, BH, a3_map(), ln_wg, ln_d, ln_part);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

}  // namespace

#ifdef FFM_ATTN3_STAMPS
extern "C" int ffm_attn3_read_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(a3_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// Entry points for attention.hip's dispatcher (same library, not part of the C ABI): FFM_EUNSUP = not this kernel's shape.
int ffm_attn3_fwd(const void* qkv, void* out, float* lse, int B, int L, int heads, int dtype, hipStream_t s) {
    if (L <= 64 || L > 256) return FFM_EUNSUP;
#define FWD3(T)                                                          \
    switch ((L + 31) / 32) {                                             \
        case 3: return run_fwd3<T, 3>(qkv, out, lse, B, L, heads, s);    \
        case 4: return run_fwd3<T, 4>(qkv, out, lse, B, L, heads, s);    \
        case 5: return run_fwd3<T, 5>(qkv, out, lse, B, L, heads, s);    \
        case 6: return run_fwd3<T, 6>(qkv, out, lse, B, L, heads, s);    \
        case 7: return run_fwd3<T, 7>(qkv, out, lse, B, L, heads, s);    \
        case 8: return run_fwd3<T, 8>(qkv, out, lse, B, L, heads, s);    \
    }
    if (dtype == FFM_BF16) { FWD3(bf16_t) }
    if (dtype == FFM_F16) { FWD3(f16_t) }
#undef FWD3
    return FFM_EUNSUP;
}
int ffm_attn3_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L, int heads,
                  int dtype, hipStream_t s, const float* ln_wg, const float* ln_d, float* ln_part) {
    if (L <= 64 || L > 256) return FFM_EUNSUP;
#define BWD3(T)                                                                               \
    switch ((L + 31) / 32) {                                                                  \
        case 3: return run_bwd3<T, 3>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
        case 4: return run_bwd3<T, 4>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
        case 5: return run_bwd3<T, 5>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
        case 6: return run_bwd3<T, 6>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
        case 7: return run_bwd3<T, 7>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
        case 8: return run_bwd3<T, 8>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s, ln_wg, ln_d, ln_part);      \
    }
    if (dtype == FFM_BF16) { BWD3(bf16_t) }
    if (dtype == FFM_F16) { BWD3(f16_t) }
#undef BWD3
    return FFM_EUNSUP;
}
