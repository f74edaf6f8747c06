// C = epilogue(A * B^T): 128x128 block tile, 4 waves of 64x64, MFMA 16x16,
// global->LDS by global_load_lds (16 B/lane, 1 KiB per wave-instruction), two
// LDS buffers, XOR-swizzled rows, fused FairLoRA / bias / residual / QuickGELU
// epilogue staged through LDS so that every global access of the epilogue is a
// full 16-byte-per-lane row segment.
//
// Byte view of a K-tile: each operand row contributes 128 bytes (64 bf16 or 32
// f32), i.e. 8 chunks of 16 B.  LDS holds row r's chunk c at slot (c ^ (r & 7));
// global_load_lds writes LDS linearly in lane order, so the swizzle is applied
// to the per-lane SOURCE address (lane -> row r = l>>3, slot p = l&7 reads
// global chunk p ^ (r&7)) and again on the ds_read side.
#include "gemm_panel.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, KT_BYTES = 128;
constexpr int TILE_BYTES = BM * KT_BYTES;            // 16 KiB per operand per buffer
constexpr int RK_ROWS = 16;                          // rows of the packed rank operand (rank padded to 16)
constexpr int RK_BYTES = RK_ROWS * KT_BYTES;         // 2 KiB per buffer
constexpr int CS_LD = 132;                           // padded f32 row of the C stage
constexpr int CS_ROWS = 64;                          // epilogue runs in two 64-row halves

// FFM_EPI_RANKOP: the rank-r down projection of FairLoRA, t = A_rows . rk^T (x A in the forward pass,
// g B^T in the backward pass), is computed INSIDE this GEMM: the packed operand rk [16, K] rides through
// the same LDS ring as 16 extra B rows and costs 2 extra MFMAs per wave per k-step (+6 %), instead of a
// separate kernel that re-reads the whole activation matrix from HBM.  ts = scaling * t * s_b is formed
// in the epilogue and used by the rank-r update of the same tile; the blocks of the first column of tiles
// also store t / ts and the dS partial sums.
template <bool RK> __host__ __device__ constexpr int buf_bytes() { return 2 * TILE_BYTES + (RK ? RK_BYTES : 0); }

// LDS = [ring: 2 buffers, reused by the epilogue for the C stage / Ts / t tile]
//       [persistent: LoRA matrix tile Ls[r][128], bias[128], ts rows [128][r] (non-RANKOP), lora_S [256] and the
//        128 rows' group ids (RANKOP)] -- filled at kernel start so that their global-load latency hides behind
//        the main loop instead of sitting in front of the epilogue.
__host__ __device__ constexpr int epi_lds_bytes(int r, bool rk) {
    return CS_ROWS * CS_LD * 4 + CS_ROWS * ((r + 7) & ~7) * 4 + (rk ? BM * RK_ROWS * 4 : 0) + (r ? BM * 64 : 0);
}
__host__ __device__ constexpr int persist_bytes(int r, bool rk) {
    const int ls = r * BN * 4 > BN * 64 ? r * BN * 4 : (r ? BN * 64 : 0);   // Ls [r][128] f32, or LwB [128][64 B]
    return ls + BN * 4 + (rk ? 256 * 4 + BM * 4 + 272 * 4 : BM * r * 4);    // (rk: lora_S, group ids, the s_b table)
}

template <typename T>
__device__ __forceinline__ void stage_tile(const T* __restrict__ g, int ld, int row0, int nrows_total,
                                           int kbyte0, char* lds_tile, int wave, int lane) {
    // 16 wave-instructions of 1 KiB (8 rows x 128 B); wave w issues 4 of them.
    const char* gb = reinterpret_cast<const char*>(g);
    const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int inst = wave * 4 + q;
        const int row = inst * 8 + rsub;              // row within the tile; (row & 7) == rsub
        int grow = row0 + row;
        grow = grow < nrows_total ? grow : nrows_total - 1;   // clamp (results of clamped rows are dropped)
        const int chunk = slot ^ rsub;
        const char* src = gb + ((size_t)grow * (size_t)ld) * sizeof(T) + kbyte0 + chunk * 16;
        char* dst = lds_tile + inst * 1024;           // wave-uniform base; hardware adds lane*16
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// Barrier for LDS hand-offs only: __syncthreads() also drains vmcnt, i.e. waits for every global STORE issued before it
// (the t / ts rows, the previous half's output rows); the epilogue's barriers order nothing but LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Kernel argument: the public ffm_gemm_args plus the implicit-convolution view of the A operand (CV instantiation).
struct gemm_kargs {
    ffm_gemm_args g;
    int conv_h, conv_w, conv_c;      // A = NHWC activation [B*H*W, C]; logical A row = the 3x3 / pad 1 patch of the pixel
    const void* conv_zero;           // >= 16 zero bytes in device memory: source of padded / out-of-image chunks
    int ksplit;                      // > 1: blockIdx.y owns a slice of the K tiles and writes an fp32 partial tile
    float* part;                     // [ksplit][M][N] fp32 partial products (summed by splitk_reduce_kernel)
};

// One 1 KiB LDS-DMA piece issued from inline asm (M0 = LDS destination, saved and restored in the same statement): hipcc
// does not see it, so it places no vmcnt(0) in front of the ds_reads that follow - the four-stage ring's waits are counted by hand.
__device__ __forceinline__ void lds_dma16(const char* src, char* dst) {
    typedef __attribute__((address_space(3))) char lds_c;
    const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_c*)dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(d)
                 : "memory");
}

// Implicit im2col: the 16-byte chunk a lane fetches for K-tile kt lies in ONE tap (C % chunk == 0); its source is the
// same channel offset of the neighbouring pixel, or the zero buffer outside the image / beyond tap 8 (K padding).
// The lane's chunk index (slot ^ rsub) does not depend on the instruction, so (tap, channel) advance once per K-tile.
template <typename T>
struct ConvA {
    int y[4], x[4], pix[4];          // the lane's four tile rows: pixel coordinates and linear pixel index
    int tap, cb;                     // tap and channel BYTE offset of the lane's chunk in the current K-tile
    __device__ __forceinline__ void init(const gemm_kargs& k, int m0, int wave, int lane, int kt0) {
        const int rsub = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int g = m0 + (wave * 4 + q) * 8 + rsub;
            g = g < k.g.M ? g : k.g.M - 1;
            pix[q] = g;
            x[q] = g % k.conv_w;
            y[q] = (g / k.conv_w) % k.conv_h;
        }
        const int kb = kt0 * KT_BYTES + (slot ^ rsub) * 16, cbytes = k.conv_c * (int)sizeof(T);
        tap = kb / cbytes;
        cb = kb % cbytes;
    }
    template <bool ASM = false>
    __device__ __forceinline__ void stage(const gemm_kargs& k, char* lds_tile, int wave) {
        const char* xb = reinterpret_cast<const char*>(k.g.a);
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const size_t cbytes = (size_t)k.conv_c * sizeof(T);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = y[q] + dy, xx = x[q] + dx;
            const bool in = tap < 9 && yy >= 0 && yy < k.conv_h && xx >= 0 && xx < k.conv_w;
            const char* src = in ? xb + (size_t)(pix[q] + dy * k.conv_w + dx) * cbytes + cb
                                 : reinterpret_cast<const char*>(k.conv_zero);
            if constexpr (ASM)
                lds_dma16(src, lds_tile + (wave * 4 + q) * 1024);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds_tile + (wave * 4 + q) * 1024),
                                                 16, 0, 0);
        }
        cb += KT_BYTES;                                          // next K-tile
        const int cbi = k.conv_c * (int)sizeof(T);
        while (cb >= cbi) { cb -= cbi; ++tap; }
    }
};

// FL >= 0: the epilogue flags are a compile-time constant (the combinations the engine uses are
// instantiated below, so the epilogue is straight-line code); FL < 0: generic, flags read at run time.
// VL: the rank-r update on the VALU (rank > 16, generic flags only).  Its 64-register LoRA tile pushes every instantiation
// that carries it into scratch (30 spilled registers in the epilogue, +8 us per launch on RN50's short-K products), so
// the rank <= 16 variants - the MFMA update - are compiled without it.
// NST: stages of the operand ring.  2 (default): the next K tile lands while this one is multiplied, two blocks per CU hide
// the rest.  4 (round 5; plain operands only): for launches of FEWER tiles than CUs - RN50's layer3 / layer4 products, 52-208
// blocks - where a block has its CU to itself and every K step waited out one LDS-DMA round trip (~1 us for 0.13 us of
// MFMAs: 34 us at K = 2048): three K tiles in flight behind hand-counted s_waitcnt vmcnt + one raw barrier per K step.
// The DMA of that path is issued from inline asm (hipcc must not see it: it would drain vmcnt(0) in front of every ds_read).
template <typename T, bool RK, int FL, bool CV = false, bool VL = false, int NST = 2>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(gemm_kargs px) {
    static_assert(NST == 2 || !VL, "the deep ring has no VALU rank-r variant");
    const ffm_gemm_args& p = px.g;
    typedef typename Mma16<T>::frag_t frag_t;
    constexpr int BUF = buf_bytes<RK>();
    const int flags = FL >= 0 ? FL : p.flags;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int logical = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = logical / tiles_n, tn = logical % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* A = reinterpret_cast<const T*>(p.a);
    const T* B = reinterpret_cast<const T*>(p.b);
    const int nk_all = (int)((size_t)p.K * sizeof(T) / KT_BYTES);
    int kt0 = 0, nk = nk_all;                                       // this block's K tiles: [kt0, kt0 + nk)
    if constexpr (CV) {
        if (px.ksplit > 1) {
            const int per = (nk_all + px.ksplit - 1) / px.ksplit;
            kt0 = blockIdx.y * per;
            nk = nk_all - kt0 < per ? nk_all - kt0 : per;
        }
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 tacc[2];
    tacc[0] = tacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the packed rank operand: 16 rows x 128 B per K-tile = 2 wave-instructions (waves 0 and 1)
    auto stage_rank = [&](int kbyte0, char* dst) {
        if (RK && wave < 2) {
            const int rsub = lane >> 3, slot = lane & 7;
            const int row = wave * 8 + rsub;
            const char* src = reinterpret_cast<const char*>(p.rk) + (size_t)row * (size_t)p.K * sizeof(T) + kbyte0 +
                              ((slot ^ rsub) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
        }
    };

    // deep ring (NST > 2): one stage = this wave's 4 + 4 (+ 1, waves 0 / 1 under RANKOP) pieces of 1 KiB, by asm LDS-DMA
    ConvA<T> cva;
    auto stage_deep = [&](int kt, char* buf) {                       // (stages are issued in K order: ConvA advances per call)
        const int rsub = lane >> 3, slot = lane & 7, chunk = slot ^ rsub;
        const char* ab = reinterpret_cast<const char*>(A);
        const char* bb = reinterpret_cast<const char*>(B);
        if constexpr (CV) cva.template stage<true>(px, buf, wave);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int inst = wave * 4 + q, row = inst * 8 + rsub;
            int ga_ = m0 + row, gb_ = n0 + row;
            ga_ = ga_ < p.M ? ga_ : p.M - 1;
            gb_ = gb_ < p.N ? gb_ : p.N - 1;
            if constexpr (!CV)
                lds_dma16(ab + ((size_t)ga_ * (size_t)p.lda) * sizeof(T) + (size_t)kt * KT_BYTES + chunk * 16, buf + inst * 1024);
            lds_dma16(bb + ((size_t)gb_ * (size_t)p.ldb) * sizeof(T) + (size_t)(kt0 + kt) * KT_BYTES + chunk * 16,
                      buf + TILE_BYTES + inst * 1024);
        }
        if (RK && wave < 2) {
            const int row = wave * 8 + rsub;
            lds_dma16(reinterpret_cast<const char*>(p.rk) + (size_t)row * (size_t)p.K * sizeof(T) + (size_t)kt * KT_BYTES + (chunk << 4),
                      buf + 2 * TILE_BYTES + wave * 1024);
        }
    };

    // prologue: tile 0 -> buffer 0
    if constexpr (NST > 2) {
        // (issued behind the epilogue operands' loads and their vmcnt(0) below: the counted waits of the loop see nothing else)
        if constexpr (CV) cva.init(px, m0, wave, lane, kt0);
    } else if constexpr (CV) {
        cva.init(px, m0, wave, lane, kt0);
        cva.stage(px, smem, wave);
    } else {
        stage_tile<T>(A, p.lda, m0, p.M, 0, smem, wave, lane);
    }
    if constexpr (NST == 2) {
        stage_tile<T>(B, p.ldb, n0, p.N, kt0 * KT_BYTES, smem + TILE_BYTES, wave, lane);
        stage_rank(0, smem + 2 * TILE_BYTES);
    }

    // ---- persistent epilogue operands (their loads overlap the main loop)
    const bool has_lora = (flags & FFM_EPI_LORA) != 0;
    const int r = has_lora ? p.rank : 0;
    // rank <= 16: the rank-r update acc += ts . lw runs on the MFMA pipe right after the main loop (operands
    // as 64-byte rows: 32 bf16 / 16 f32 rank slots, zero padded); larger ranks use the VALU update in the epilogue
    constexpr int KE = 64 / (int)sizeof(T);
    const bool lora_mma = has_lora && !VL;
    float* Ls = reinterpret_cast<float*>(smem + NST * BUF);   // LoRA matrix tile [r][BN] (VALU path)
    T* LwB = reinterpret_cast<T*>(smem + NST * BUF);          // LoRA matrix tile, transposed [BN][KE] (MFMA path)
    const int ls_bytes = r * BN * 4 > BN * 64 ? r * BN * 4 : (r ? BN * 64 : 0);
    float* Bias = reinterpret_cast<float*>(smem + NST * BUF + ls_bytes); // [BN]
    float* TsAll = Bias + BN;                                 // non-RANKOP: ts rows [BM][r]
    float* Sg = Bias + BN;                                    // RANKOP: lora_S [G][r] (<= 256 floats)
    int* Ga = reinterpret_cast<int*>(Sg + 256);               // RANKOP: group id of each tile row (-1: uniform mix)
    float* SBt = reinterpret_cast<float*>(Ga + BM);           // RANKOP: s_b of every group mix [(G + 1)][r]: row 0 uniform, row 1 + a for group a
    // Every global load of these operands is ISSUED before the first LDS store of any of them: a load -> store -> load ->
    // store sequence pays one memory round trip per operand (bias, LoRA tile, lora_S, group ids: four in a row, and a
    // short-K product has no main loop to hide them behind).
    // ... as inline-asm loads with ONE hand-placed wait: beside the LDS-DMA fills above hipcc puts s_waitcnt vmcnt(0) in
    // front of the first use of every ordinary load, and a load under a per-lane condition ends its basic block with
    // one (ten to twenty of them in a row made this prologue 7 300 cycles against the plain kernel's 2 200).  Indices are
    // clamped, the loads sit in wave-uniform branches, out-of-range lanes are zeroed after the wait.
    float bias_v = 0.f, sg_v = 0.f, lwv[8], tsr[8], tfv[8];
    int ga_v = -1;
    auto ldg = [](const float* q) -> float {
        float v;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
        return v;
    };
    const int rsh = (r > 0 && (r & (r - 1)) == 0) ? __builtin_ctz(r) : -1;       // r a power of two: idx / r is a shift
    const bool do_ds = RK && (tn == 0) && p.t_fwd && p.ds_part;
#pragma unroll
    for (int e = 0; e < 8; ++e) lwv[e] = tsr[e] = tfv[e] = 0.f;
    if (flags & FFM_EPI_BIAS) {
        const int n = n0 + (tid & (BN - 1));
        bias_v = ldg(p.bias + (n < p.N ? n : p.N - 1));
    }
    if (lora_mma) {
        // LoRA tile (MFMA path): thread -> column n = tid & 127 and the eight rank slots [8q, 8q + 8), q = tid >> 7
        const int fn = tid & 127, q = tid >> 7;
        const int n = (n0 + fn) < p.N ? (n0 + fn) : (p.N - 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * q + e, jc = j < r ? j : r - 1;
            lwv[e] = ldg((flags & FFM_EPI_LORA_KR) ? p.lw + (size_t)n * r + jc : p.lw + (size_t)jc * p.N + n);
        }
        if constexpr (!RK) {
            // non-RANKOP: row tid & 127's ts values for the same slots stay in registers until the rank-r update
            const int gm = (m0 + fn) < p.M ? (m0 + fn) : (p.M - 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 8 * q + e, jc = j < r ? j : r - 1;
                tsr[e] = ldg(p.ts + (size_t)gm * r + jc);
            }
        }
    }
    if constexpr (RK) {
        if (has_lora) {
            const int gr = p.G * r;
            sg_v = ldg(p.S + (tid < gr ? tid : gr - 1));
            if (p.attr) {
                const int gm = (m0 + (tid & 127)) < p.M ? (m0 + (tid & 127)) : (p.M - 1);
                asm volatile("global_load_dword %0, %1, off" : "=v"(ga_v) : "v"(p.attr + gm / p.rows_per_sample) : "memory");
            }
        }
        if (do_ds) {
            // dS partials (backward, first column tile): this thread's t_fwd values, element idx = tid + 256 it of the
            // tile's [128][r] block - loaded now so that their latency hides behind the main loop
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = tid + 256 * it, ic = idx < BM * r ? idx : BM * r - 1;
                const int row = rsh >= 0 ? ic >> rsh : ic / r;
                const int gm = (m0 + row) < p.M ? (m0 + row) : (p.M - 1);
                tfv[it] = ldg(p.t_fwd + (size_t)gm * r + (ic - row * r));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(bias_v), "+v"(sg_v), "+v"(ga_v));
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(lwv[e]), "+v"(tsr[e]), "+v"(tfv[e]));
    {
        // zero what the clamps stood in for
        const int fn = tid & 127, q = tid >> 7;
        if (n0 + (tid & (BN - 1)) >= p.N) bias_v = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (8 * q + e >= r || n0 + fn >= p.N) lwv[e] = 0.f;
            if (8 * q + e >= r || m0 + fn >= p.M) tsr[e] = 0.f;
        }
        if constexpr (RK) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = tid + 256 * it;
                const int row = rsh >= 0 ? idx >> rsh : idx / (r > 0 ? r : 1);
                if (idx >= BM * r || m0 + row >= p.M) tfv[it] = 0.f;
            }
        }
    }
    // ---- the LDS stores
    if (tid < BN) Bias[tid] = bias_v;
    if (lora_mma) {
        // the tile row [KE] is written as whole 8-element segments (2-byte stores at a 64-byte lane stride were a 16-way
        // bank conflict, and the LDS pipe is shared by the CU's eight waves)
        const int fn = tid & 127, q = tid >> 7;
        Vec8<T>::store(LwB + fn * KE + 8 * q, lwv);
        if constexpr (KE > 16) {                           // slots 16 .. KE are padding
            const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            Vec8<T>::store(LwB + fn * KE + 16 + 8 * q, z);
        }
    }
    if constexpr (RK) {
        if (has_lora) {
            if (tid < p.G * r) Sg[tid] = sg_v;
            if (tid < BM) Ga[tid] = ga_v;
        }
    }
    if constexpr (VL) {
        if (has_lora) {                                    // rank > 16: fp32 LoRA tile [r][BN] and (non-RANKOP) the ts rows
            for (int idx0 = tid; idx0 < r * BN; idx0 += 256 * 8) {
                float tmp[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = idx0 + 256 * it;
                    const int j = idx / BN, n = idx % BN;
                    tmp[it] = 0.f;
                    if (idx < r * BN && n0 + n < p.N)
                        tmp[it] = (flags & FFM_EPI_LORA_KR) ? p.lw[(size_t)(n0 + n) * r + j] : p.lw[(size_t)j * p.N + n0 + n];
                }
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    if (idx0 + 256 * it < r * BN) Ls[idx0 + 256 * it] = tmp[it];
            }
            if constexpr (!RK) {
                const int frow_ = tid & 127, fj0 = tid >> 7;
                const int gm = m0 + frow_;
                for (int jb = 0; jb < r; jb += 16) {
                    float tmp[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int j = jb + fj0 + 2 * it;
                        tmp[it] = (j < r && gm < p.M) ? p.ts[(size_t)gm * r + j] : 0.f;
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int j = jb + fj0 + 2 * it;
                        if (j < r) TsAll[frow_ * r + j] = tmp[it];
                    }
                }
            }
        }
    }

    __syncthreads();
    if constexpr (RK) {
        // s_b[j] = sum_g pi_b[g] S[g][j] takes G + 1 values per rank slot (uniform mix, or the sample's group a): one
        // table entry per thread here instead of G multiply-adds per tile element in the rank-r stage (read after the main
        // loop's barriers)
        if (has_lora && tid < (p.G + 1) * r && (p.G + 1) * r <= 272) {
            const int cls = rsh >= 0 ? tid >> rsh : tid / r, j = tid - cls * r;
            const float mu = 1.0f / (float)p.G, mo = (1.0f - p.lambda_group) / (float)(p.G - 1);
            float sb = 0.f;
            for (int g = 0; g < p.G; ++g) sb += (cls == 0 ? mu : (cls - 1 == g ? p.lambda_group : mo)) * Sg[g * r + j];
            SBt[tid] = sb;
        }
    }

    const int frow = lane & 15, fgrp = lane >> 4;
    if constexpr (NST > 2) {
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0)
            if (s0 < nk) stage_deep(s0, smem + s0 * BUF);
        const bool nine = RK && wave < 2;                           // pieces per stage of this wave: 9 or 8
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed when at most the y younger stages (kt + 1 .. kt + NST - 2) are still in flight
            const int y = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
            if (nine) {
                if (y >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else if (y == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (y >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (y == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // every wave's pieces of stage kt are in LDS, and every wave is done reading stage kt - 1: its buffer is free
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + NST - 1 < nk) stage_deep(kt + NST - 1, smem + ((kt + NST - 1) % NST) * BUF);
            const char* As = smem + (kt % NST) * BUF;
            const char* Bs = As + TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int chunk = ks * 4 + fgrp;
                frag_t af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm * 64 + i * 16 + frow;
                    af[i] = *reinterpret_cast<const frag_t*>(As + ra * KT_BYTES + ((chunk ^ (ra & 7)) << 4));
                    const int rb = wn * 64 + i * 16 + frow;
                    bf[i] = *reinterpret_cast<const frag_t*>(Bs + rb * KT_BYTES + ((chunk ^ (rb & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], af[i], bf[j]);
                if constexpr (RK) {
                    const frag_t kf = *reinterpret_cast<const frag_t*>(As + 2 * TILE_BYTES + frow * KT_BYTES +
                                                                       ((chunk ^ (frow & 7)) << 4));
                    if (wn == 0) { Mma16<T>::mma(tacc[0], af[0], kf); Mma16<T>::mma(tacc[1], af[1], kf); }
                    else         { Mma16<T>::mma(tacc[0], af[2], kf); Mma16<T>::mma(tacc[1], af[3], kf); }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the ring is the epilogue's stage from here on
    }
    for (int kt = 0; NST == 2 && kt < nk; ++kt) {
        const int cur = kt & 1;
        char* As = smem + cur * BUF;
        char* Bs = As + TILE_BYTES;
        if (kt + 1 < nk) {
            char* An = smem + (cur ^ 1) * BUF;
            if constexpr (CV) cva.stage(px, An, wave);
            else stage_tile<T>(A, p.lda, m0, p.M, (kt + 1) * KT_BYTES, An, wave, lane);
            stage_tile<T>(B, p.ldb, n0, p.N, (kt0 + kt + 1) * KT_BYTES, An + TILE_BYTES, wave, lane);
            stage_rank((kt + 1) * KT_BYTES, An + 2 * TILE_BYTES);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + fgrp;
            frag_t af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + frow;
                af[i] = *reinterpret_cast<const frag_t*>(As + ra * KT_BYTES + ((chunk ^ (ra & 7)) << 4));
                const int rb = wn * 64 + i * 16 + frow;
                bf[i] = *reinterpret_cast<const frag_t*>(Bs + rb * KT_BYTES + ((chunk ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], af[i], bf[j]);
            if constexpr (RK) {
                // t tile: wave (wm, wn) owns row fragments 2*wn, 2*wn+1 of its 64 rows
                const frag_t kf = *reinterpret_cast<const frag_t*>(As + 2 * TILE_BYTES + frow * KT_BYTES +
                                                                   ((chunk ^ (frow & 7)) << 4));
                if (wn == 0) { Mma16<T>::mma(tacc[0], af[0], kf); Mma16<T>::mma(tacc[1], af[1], kf); }
                else         { Mma16<T>::mma(tacc[0], af[2], kf); Mma16<T>::mma(tacc[1], af[3], kf); }
            }
        }
        __syncthreads();   // next tile landed (compiler drains vmcnt before the barrier); cur is free
    }

    if constexpr (CV) {
        if (px.ksplit > 1) {                                        // fp32 partial tile, summed by splitk_reduce_kernel
            float* P = px.part + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = m0 + wm * 64 + i * 16 + fgrp * 4 + e, col = n0 + wn * 64 + j * 16 + frow;
                        if (row < p.M && col < p.N) P[(size_t)row * p.N + col] = acc[i][j][e];
                    }
            return;
        }
    }

    // ---------------- epilogue: two halves of 64 rows through LDS ----------
    float* Cs = reinterpret_cast<float*>(smem);
    const int rp8 = (r + 7) & ~7;                     // Ts row stride: rank padded to 8 with zeros (branch-free update)
    float* Ts = Cs + CS_ROWS * CS_LD;                 // ts rows of the current half [64][rp8]
    float* Tt = Ts + CS_ROWS * rp8;                   // RANKOP: t tile [128][16]
    T* C = reinterpret_cast<T*>(p.c);
    const float mix_u = 1.0f / (float)p.G, mix_o = (1.0f - p.lambda_group) / (float)(p.G - 1);
    auto mixw = [&](int row, int g) -> float {        // pi_b[g] of the sample that owns tile row `row`
        const int a = Ga[row];
        return a < 0 ? mix_u : (a == g ? p.lambda_group : mix_o);
    };
    if constexpr (RK) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Tt[(wm * 64 + (2 * wn + ii) * 16 + fgrp * 4 + e) * RK_ROWS + frow] = tacc[ii][e];
        lds_barrier();                              // Tt complete
        if (do_ds) {
            // dS partial of this row tile: sum_rows pi_b[g] * scaling * t_fwd * t.  Element idx = tid + 256 it is
            // (row idx / r, slot idx % r); with r a power of two a thread keeps ONE slot, so its rows add up in registers,
            // the lanes of a slot by shuffles, the four waves through LDS (any other r: one thread per (g, slot))
            float* Red = Cs;                          // [4][G * r]
            if ((r & (r - 1)) == 0) {
                // (branch-free: tile row, t and group id are read with clamped indices, invalid elements weigh zero)
                float wv[8];
                int gaw[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = tid + 256 * it;
                    const bool ok = idx < BM * r;
                    const int rw = ok ? idx >> rsh : 0;
                    const float tt = Tt[rw * RK_ROWS + (idx & (r - 1))];
                    wv[it] = ok ? p.scaling * tfv[it] * tt : 0.f;
                    gaw[it] = Ga[rw];
                }
                for (int g = 0; g < p.G; ++g) {
                    float sacc = 0.f;
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        sacc += (gaw[it] < 0 ? mix_u : (gaw[it] == g ? p.lambda_group : mix_o)) * wv[it];
                    for (int o = r; o < 64; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
                    if (lane < r) Red[wave * (p.G * r) + g * r + lane] = sacc;
                }
                lds_barrier();
                if (tid < p.G * r)
                    p.ds_part[(size_t)tm * p.G * r + tid] = (Red[tid] + Red[p.G * r + tid]) + (Red[2 * p.G * r + tid] + Red[3 * p.G * r + tid]);
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = tid + 256 * it;
                    if (idx < BM * r) Cs[idx] = p.scaling * tfv[it] * Tt[(idx / r) * RK_ROWS + idx % r];
                }
                lds_barrier();
                if (tid < p.G * r) {
                    const int g = tid / r, j = tid % r;
                    float sacc = 0.f;
                    for (int row = 0; row < BM; ++row) sacc += mixw(row, g) * Cs[row * r + j];
                    p.ds_part[((size_t)tm * p.G + g) * r + j] = sacc;
                }
            }
            lds_barrier();                          // Cs is reused by the halves below
        }
    }
    if (lora_mma) {
        T* TsA = reinterpret_cast<T*>(Tt + (RK ? BM * RK_ROWS : 0));      // ts tile [BM][KE], zero padded
        {
            // thread -> tile row tid & 127, rank slots [8q, 8q + 8): one 8-element segment per thread (+ one of padding)
            const int row = tid & 127, q = tid >> 7;
            const int gm = m0 + row;
            const bool sb_tab = (p.G + 1) * r <= 272;
            const bool vec_out = RK && (r & 3) == 0;          // t / ts rows leave as 16-byte stores
            float tsv[8], tvv[8];
            if constexpr (RK) {
                // branch-free: the t row, the group id and the s_b table row are read unconditionally (a read under a
                // per-lane condition is a basic block of its own that ends in s_waitcnt lgkmcnt(0): 24 LDS round trips in
                // a row), the slots beyond r and the rows beyond M are zeroed afterwards
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(&Tt[row * RK_ROWS + 8 * q]);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(&Tt[row * RK_ROWS + 8 * q + 4]);
                float sbv[8];
                if (sb_tab) {
                    const int base = (Ga[row] + 1) * r + 8 * q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sbv[e] = SBt[base + e < 272 ? base + e : 271];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = 8 * q + e;
                        float sb = 0.f;
                        if (j < r) for (int g = 0; g < p.G; ++g) sb += mixw(row, g) * Sg[g * r + j];
                        sbv[e] = sb;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = 8 * q + e < r && gm < p.M;
                    const float tv = e < 4 ? t0[e & 3] : t1[e & 3];
                    tvv[e] = ok ? tv : 0.f;
                    tsv[e] = ok ? p.scaling * tv * sbv[e] : 0.f;
                }
                if (tn == 0 && !vec_out && gm < p.M) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = 8 * q + e;
                        if (j < r) {
                            if (p.t_out) p.t_out[(size_t)gm * r + j] = tvv[e];
                            if (p.ts_out) p.ts_out[(size_t)gm * r + j] = tsv[e];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) tsv[e] = tsr[e];
            }
            if constexpr (RK) {
                if (vec_out && tn == 0 && gm < p.M) {
#pragma unroll
                    for (int e0 = 0; e0 < 8; e0 += 4) {
                        if (8 * q + e0 < r) {
                            if (p.t_out) *reinterpret_cast<f32x4*>(p.t_out + (size_t)gm * r + 8 * q + e0) = (f32x4){tvv[e0], tvv[e0 + 1], tvv[e0 + 2], tvv[e0 + 3]};
                            if (p.ts_out) *reinterpret_cast<f32x4*>(p.ts_out + (size_t)gm * r + 8 * q + e0) = (f32x4){tsv[e0], tsv[e0 + 1], tsv[e0 + 2], tsv[e0 + 3]};
                        }
                    }
                }
            }
            Vec8<T>::store(TsA + row * KE + 8 * q, tsv);
            if constexpr (KE > 16) {
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                Vec8<T>::store(TsA + row * KE + 16 + 8 * q, z);
            }
        }
        lds_barrier();
        const char* ta = reinterpret_cast<const char*>(TsA);
        const char* lb = reinterpret_cast<const char*>(LwB);
        frag_t bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const frag_t*>(lb + (wn * 64 + j * 16 + frow) * 64 + fgrp * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const frag_t afr = *reinterpret_cast<const frag_t*>(ta + (wm * 64 + i * 16 + frow) * 64 + fgrp * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma16<T>::mma(acc[i][j], afr, bfr[j]);
        }
    }

    const int ecol = (tid & 15) * 8;                  // this thread's 8 columns
    const int erow0 = tid >> 4;                       // rows erow0 + 16*i
    // colstat_part: column sums of the stored tile (the BatchNorm statistics of RN50's convolutions)
    const bool cst = p.colstat_part != nullptr;
    // FFM_EPI_BNBWD only in the kernels instantiated with the bit (a run-time branch cost every kernel of this file ~60
    // registers: 134 -> 193 in the plain one)
    constexpr bool BNB = FL > 0 && (FL & FFM_EPI_BNBWD) != 0;
    const bool bnb = BNB && cst && p.bn_x != nullptr;
    float cs[8], cq[8], bnmu[8], bnrs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = cq[c] = bnmu[c] = bnrs[c] = 0.f;
    if (bnb && n0 + ecol < p.N) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { bnmu[c] = p.bn_mean[n0 + ecol + c]; bnrs[c] = p.bn_rstd[n0 + ecol + c]; }
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // this half's residual / pre-activation rows: issue the global loads now, consume after the barrier
        constexpr int NCH = (int)(sizeof(T) * 8 / 16);               // 16-byte chunks per 8 elements (1 bf16, 2 f32)
        typename Elem<T>::chunk_t rres[4][NCH], raux[4][NCH];
        const int gn = n0 + ecol;
        // (inline-asm loads with clamped addresses, one wait behind the barrier: as compiler-visible loads under the
        // per-lane bounds test each of them - and then each of the four stores that use them - sat behind its own
        // s_waitcnt vmcnt(0): eight serialised memory round trips per half tile)
        const bool has_pre = (flags & (FFM_EPI_RESIDUAL | FFM_EPI_DGELU)) != 0;
        if (has_pre) {
            const int gnc = gn < p.N ? gn : (p.N >= 8 ? p.N - 8 : 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm0 = m0 + half * 64 + erow0 + 16 * i;
                const size_t off = (size_t)(gm0 < p.M ? gm0 : p.M - 1) * p.ldc + gnc;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (flags & FFM_EPI_RESIDUAL)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rres[i][c]) : "v"(reinterpret_cast<const typename Elem<T>::chunk_t*>(reinterpret_cast<const T*>(p.res) + off) + c) : "memory");
                    if (flags & FFM_EPI_DGELU)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raux[i][c]) : "v"(reinterpret_cast<const typename Elem<T>::chunk_t*>(reinterpret_cast<const T*>(p.aux) + off) + c) : "memory");
                }
            }
        }
        // FFM_EPI_BNBWD: the BatchNorm's input and ReLU-output rows of this half, the same way
        typename Elem<T>::chunk_t rbx[BNB ? 4 : 1][NCH], rbm[BNB ? 4 : 1][NCH];
        if constexpr (BNB) if (bnb) {
            const int gnc = gn < p.N ? gn : (p.N >= 8 ? p.N - 8 : 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm0 = m0 + half * 64 + erow0 + 16 * i;
                const size_t off = (size_t)(gm0 < p.M ? gm0 : p.M - 1) * p.ldc + gnc;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rbx[i][c]) : "v"(reinterpret_cast<const typename Elem<T>::chunk_t*>(reinterpret_cast<const T*>(p.bn_x) + off) + c) : "memory");
                    if (p.bn_mask)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rbm[i][c]) : "v"(reinterpret_cast<const typename Elem<T>::chunk_t*>(reinterpret_cast<const T*>(p.bn_mask) + off) + c) : "memory");
                }
            }
        }
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Cs[(i * 16 + fgrp * 4 + e) * CS_LD + wn * 64 + j * 16 + frow] = acc[i][j][e];
        }
        if (has_lora && !lora_mma) {
            for (int idx = tid; idx < CS_ROWS * rp8; idx += 256) {
                const int row = idx / rp8, j = idx % rp8;
                const int gm = m0 + half * 64 + row;
                if (j >= r) { Ts[idx] = 0.f; continue; }
                if constexpr (RK) {
                    float tsv = 0.f;
                    if (gm < p.M) {
                        const float tv = Tt[(half * 64 + row) * RK_ROWS + j];
                        float sb = 0.f;
                        for (int g = 0; g < p.G; ++g) sb += mixw(half * 64 + row, g) * Sg[g * r + j];
                        tsv = p.scaling * tv * sb;
                        if (tn == 0) {
                            if (p.t_out) p.t_out[(size_t)gm * r + j] = tv;
                            if (p.ts_out) p.ts_out[(size_t)gm * r + j] = tsv;
                        }
                    }
                    Ts[idx] = tsv;
                } else {
                    Ts[idx] = TsAll[(half * 64 + row) * r + j];
                }
            }
        }
        lds_barrier();
        if (has_pre) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (flags & FFM_EPI_RESIDUAL) asm volatile("" : "+v"(rres[i][c]));
                    if (flags & FFM_EPI_DGELU) asm volatile("" : "+v"(raux[i][c]));
                }
        }
        if constexpr (BNB) if (bnb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    asm volatile("" : "+v"(rbx[i][c]));
                    if (p.bn_mask) asm volatile("" : "+v"(rbm[i][c]));
                }
        }
        if (gn < p.N) {
            float v[4][8];
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(&Bias[ecol]);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(&Bias[ecol + 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int lrow = erow0 + 16 * i;
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(&Cs[lrow * CS_LD + ecol]);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(&Cs[lrow * CS_LD + ecol + 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { v[i][c] = c0[c] + b0[c]; v[i][4 + c] = c1[c] + b1[c]; }
                }
            }
            if constexpr (VL) if (has_lora) {
                // rank-r update (rank > 16): this thread's 8 columns of the LoRA matrix stay in registers
                for (int j0 = 0; j0 < r; j0 += 8) {
                    float lreg[8][8];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = (j0 + jj) < r ? (j0 + jj) : (r - 1);
                        const f32x4 l0 = *reinterpret_cast<const f32x4*>(&Ls[j * BN + ecol]);
                        const f32x4 l1 = *reinterpret_cast<const f32x4*>(&Ls[j * BN + ecol + 4]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) { lreg[jj][c] = l0[c]; lreg[jj][4 + c] = l1[c]; }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int lrow = erow0 + 16 * i;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const float tj = Ts[lrow * rp8 + j0 + jj];
#pragma unroll
                            for (int c = 0; c < 8; ++c) v[i][c] += tj * lreg[jj][c];
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lrow = erow0 + 16 * i;
                const int gm = m0 + half * 64 + lrow;
                if (gm >= p.M) continue;
                const size_t off = (size_t)gm * p.ldc + gn;
                constexpr int EPC = Elem<T>::kPerChunk;
                if (flags & FFM_EPI_RESIDUAL) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[i][c] += Elem<T>::to_f(rres[i][c / EPC][c % EPC]);
                }
                if (flags & FFM_EPI_DGELU) {
                    if (p.gelu_deriv) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[i][c] *= Elem<T>::to_f(raux[i][c / EPC][c % EPC]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[i][c] *= Act<T>::gelu_grad(Elem<T>::to_f(raux[i][c / EPC][c % EPC]));
                    }
                }
                float gd[8];
                const bool deriv = (flags & FFM_EPI_GELU) && p.gelu_deriv;
                if (deriv) {
                    float ga[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) Act<T>::gelu_both(Elem<T>::to_f(Elem<T>::from_f(v[i][c])), ga[c], gd[c]);
                    Vec8<T>::store(C + off, gd);
                    Vec8<T>::store(reinterpret_cast<T*>(p.c2) + off, ga);
                } else {
                    Vec8<T>::store(C + off, v[i]);
                }
                if (BNB && cst && bnb) {
                    // {sum g, sum g xhat}, g = stored value * (ReLU output > 0): ffm_bn_bwd's column sums (colsum_kernel MODE 1)
                    float gm[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float g = Elem<T>::to_f(Elem<T>::from_f(v[i][c]));
                        if (p.bn_mask && !(Elem<T>::to_f(rbm[BNB ? i : 0][c / EPC][c % EPC]) > 0.f)) g = 0.f;
                        cs[c] += g;
                        cq[c] += g * (Elem<T>::to_f(rbx[BNB ? i : 0][c / EPC][c % EPC]) - bnmu[c]) * bnrs[c];
                        gm[c] = g;
                    }
                    if (p.bn_gout) Vec8<T>::store(reinterpret_cast<T*>(p.bn_gout) + off, gm);   // (ffm_bn_bwd's g_out)
                } else if (cst) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float st = Elem<T>::to_f(Elem<T>::from_f(v[i][c]));
                        cs[c] += st;
                        cq[c] += st * st;
                    }
                }
                if ((flags & FFM_EPI_GELU) && !deriv) {
                    float a[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) a[c] = Act<T>::gelu(Elem<T>::to_f(Elem<T>::from_f(v[i][c])));
                    Vec8<T>::store(reinterpret_cast<T*>(p.c2) + off, a);
                }
            }
        }
        lds_barrier();
    }
    if (cst) {
        // the 16 row lanes of a column chunk meet in LDS and are added in a fixed order: part[tm][0 / 1][n0 + col]
        float* R0 = Cs;
        float* R1 = Cs + 16 * BN;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            R0[erow0 * BN + ecol + c] = cs[c];
            R1[erow0 * BN + ecol + c] = cq[c];
        }
        lds_barrier();
        if (tid < BN && n0 + tid < p.N) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                s0 += R0[l * BN + tid];
                s1 += R1[l * BN + tid];
            }
            p.colstat_part[((size_t)tm * 2) * p.N + n0 + tid] = s0;
            p.colstat_part[((size_t)tm * 2 + 1) * p.N + n0 + tid] = s1;
        }
    }
}

template <typename T, bool RK, int FL, bool CV = false, bool VL = false, int NST = 2>
int launch_gemm(const ffm_gemm_args& a, hipStream_t s, const gemm_kargs* conv = nullptr) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int r = (a.flags & FFM_EPI_LORA) ? a.rank : 0;
    int lds = epi_lds_bytes(r, RK);
    if (lds < NST * buf_bytes<RK>()) lds = NST * buf_bytes<RK>();
    lds += persist_bytes(r, RK);
    if (lds > 160 * 1024) return FFM_EUNSUP;
    if (lds > 65536) {
        static bool done = false;                     // one per instantiation
        if (!done) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<T, RK, FL, CV, VL, NST>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, NST == 2 ? 80 * 1024 : 160 * 1024);
            if (e != hipSuccess) return (int)e;
            done = true;
        }
    }
    gemm_kargs ka;
    if (conv) ka = *conv;
    else { ka.g = a; ka.conv_h = ka.conv_w = ka.conv_c = 0; ka.conv_zero = nullptr; ka.ksplit = 1; ka.part = nullptr; }
    hipLaunchKernelGGL((gemm_nt_kernel<T, RK, FL, CV, VL, NST>), dim3(tiles, ka.ksplit > 1 ? ka.ksplit : 1), dim3(256), lds, s, ka);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// packed rank operands: dst[j][k] (dtype, 16 rows, rows >= r zero) from lora_A [K, r] or lora_B [r, K]
template <typename T>
__global__ __launch_bounds__(256) void lora_pack_kernel(const ffm_pack_desc* __restrict__ descs) {
    const ffm_pack_desc d = descs[blockIdx.y];
    const int total = RK_ROWS * d.K;
    T* dst = reinterpret_cast<T*>(d.dst);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i / d.K, k = i % d.K;
        float v = 0.f;
        if (j < d.r) v = d.layout_rk ? d.src[(size_t)j * d.K + k] : d.src[(size_t)k * d.r + j];
        if (d.gamma) v *= d.gamma[k];                      // LayerNorm folded into the product (ffm_gemm_args.ln_rk)
        if (j == 14 && d.row14) v = d.row14[k];            // two caller vectors beside the r <= 14 rank rows (ffm_pack_desc.row14)
        if (j == 15 && d.row15) v = d.row15[k];
        dst[i] = Elem<T>::from_f(v);
    }
    if (d.dst_wide) {                                      // [K][32]: the `lw` tile form of the panel GEMM's rank-r update
        T* wide = reinterpret_cast<T*>(d.dst_wide);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * d.K; i += gridDim.x * blockDim.x) {
            const int k = i >> 5, j = i & 31;
            float v = 0.f;
            if (j < d.r) v = d.layout_rk ? d.src[(size_t)j * d.K + k] : d.src[(size_t)k * d.r + j];
            wide[i] = Elem<T>::from_f(v);
        }
    }
}

// ln_rk[j] = sum_k dst[j][k] (the gamma-scaled operand as rounded), ln_rk[16 + j] = sum_k beta[k] lora_A[k][j]:
// one block per descriptor, 16 lanes per rank slot, fixed order
template <typename T>
__global__ __launch_bounds__(256) void lora_pack_ln_kernel(const ffm_pack_desc* __restrict__ descs) {
    const ffm_pack_desc d = descs[blockIdx.x];
    if (!d.ln_rk || !d.gamma || !d.beta || d.layout_rk) return;
    const int j = threadIdx.x >> 4, q = threadIdx.x & 15;
    const T* dst = reinterpret_cast<const T*>(d.dst);
    float c = 0.f, b = 0.f;
    for (int k = q; k < d.K; k += 16) {
        c += Elem<T>::to_f(dst[(size_t)j * d.K + k]);
        if (j < d.r) b += d.beta[k] * d.src[(size_t)k * d.r + j];
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        c += __shfl_xor(c, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    if (q == 0) {
        d.ln_rk[j] = c;
        d.ln_rk[16 + j] = b;
    }
}

}  // namespace

extern "C" int ffm_gemm_nt(const ffm_gemm_args* args, int dtype, void* stream) {
    if (!args || !args->a || !args->b || !args->c) return FFM_EINVAL;
    const ffm_gemm_args& a = *args;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    const size_t esb = dtype == FFM_F32_X3_W16 ? 2 : es;                  // (FFM_F32_X3_W16: f32 activations, half weights)
    if (dtype != FFM_BF16 && dtype != FFM_F32 && dtype != FFM_F32_X3 && dtype != FFM_F32_X3_W16) return FFM_EINVAL;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return FFM_EINVAL;
    if (((size_t)a.K * esb) % KT_BYTES != 0 || a.N % 8 != 0) return FFM_EINVAL;
    if (((size_t)a.lda * es) % 16 || ((size_t)a.ldb * esb) % 16 || ((size_t)a.ldc * es) % 16) return FFM_EINVAL;
    if (((uintptr_t)a.a | (uintptr_t)a.b | (uintptr_t)a.c) & 15) return FFM_EINVAL;
    if (a.lda < a.K || a.ldb < a.K || a.ldc < a.N) return FFM_EINVAL;
    const bool rk = (a.flags & FFM_EPI_RANKOP) != 0;
    if ((a.flags & FFM_EPI_LORA) && (a.rank <= 0 || a.rank > FFM_MAX_RANK || !a.lw || (!rk && !a.ts))) return FFM_EINVAL;
    if (rk) {
        if (!(a.flags & FFM_EPI_LORA) || a.rank > RK_ROWS || !a.rk || ((uintptr_t)a.rk & 15) || !a.S) return FFM_EINVAL;
        if (a.G <= 0 || a.G > FFM_MAX_GROUPS || a.rows_per_sample <= 0 || a.G * a.rank > 256) return FFM_EINVAL;
        if ((a.t_fwd == nullptr) != (a.ds_part == nullptr)) return FFM_EINVAL;
    }
    if ((a.flags & FFM_EPI_BIAS) && !a.bias) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_RESIDUAL) && (!a.res || ((uintptr_t)a.res & 15))) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_GELU) && (!a.c2 || ((uintptr_t)a.c2 & 15))) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_DGELU) && (!a.aux || ((uintptr_t)a.aux & 15))) return FFM_EINVAL;
    if (a.flags & FFM_EPI_BNBWD) {                                        // BatchNorm-backward column sums: 128x128 kernel
        if (!a.colstat_part || !a.bn_x || !a.bn_mean || !a.bn_rstd || (a.flags & FFM_EPI_GELU)) return FFM_EINVAL;
        if (((uintptr_t)a.bn_x | (uintptr_t)a.bn_mask | (uintptr_t)a.bn_gout) & 15) return FFM_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (ffm_skinny_ok(a, dtype) && !a.colstat_part) {                    // (column sums: the 128x128 kernel's epilogue)
        static const bool off = getenv("FFM_SKINNY") && getenv("FFM_SKINNY")[0] == 'o';      // FFM_SKINNY=off: A/B runs
        if (!off || dtype == FFM_F32_X3 || dtype == FFM_F32_X3_W16) return ffm_skinny_launch(a, dtype, s);
    }
    if (dtype == FFM_F32_X3 || dtype == FFM_F32_X3_W16) return FFM_EUNSUP;      // split-operand products: skinny shapes only
    if (a.flags & (FFM_EPI_LNB_STAT | FFM_EPI_LNB_APPLY)) {              // LayerNorm backward folded in: one panel tile each
        if ((a.flags & FFM_EPI_LNB_STAT) && !a.lnb_part) return FFM_EINVAL;
        if ((a.flags & FFM_EPI_LNB_APPLY) && rk && a.rank > 14) return FFM_EUNSUP;   // rows 14 / 15 of rk carry W gamma and d
        if ((a.flags & FFM_EPI_LNB_APPLY) && (!a.lnb_part || a.lnb_np <= 0 || a.lnb_np > (rk ? 8 : 24) || !a.lnb_x || ((uintptr_t)a.lnb_x & 15) ||
                                             !a.lnb_gamma || !a.ln_mean || !a.ln_rstd || (rk && !a.ln_rk) || !a.res || ((uintptr_t)a.res & 15)))
            return FFM_EINVAL;
        const int cfgn = (a.b_packed && !a.colstat_part) ? ffm_panel_select(a.M, a.N, a.K, a.flags, a.rank, dtype, true) : -1;
        return cfgn >= 0 ? ffm_panel_launch(a, cfgn, s) : FFM_EUNSUP;
    }
    if (a.flags & FFM_EPI_LGRAD) {                                        // gradient partial products: one panel tile only
        const int cfgg = (a.b_packed && !a.colstat_part) ? ffm_panel_select(a.M, a.N, a.K, a.flags, a.rank, dtype, true) : -1;
        return cfgg >= 0 ? ffm_panel_launch(a, cfgg, s) : FFM_EUNSUP;
    }
    if (a.flags & (FFM_EPI_ROWSTATS | FFM_EPI_LNIN)) {                    // LayerNorm folding: the panel kernel only
        if (a.colstat_part) return FFM_EUNSUP;                            // ... which has no column-sum epilogue
        const int cfgl = a.b_packed ? ffm_panel_select(a.M, a.N, a.K, a.flags, a.rank, dtype, true) : -1;
        return cfgl >= 0 ? ffm_panel_launch(a, cfgl, s) : FFM_EUNSUP;
    }
    // column sums (colstat_part) are an epilogue of the 128x128 / 128xN kernels only: a packed weight does not send
    // such a launch to the panel kernel, which would return without writing them
    if (a.b_packed && !a.colstat_part) {
        const int cfg = ffm_panel_select(a.M, a.N, a.K, a.flags, a.rank, dtype, true);
        if (cfg >= 0) return ffm_panel_launch(a, cfg, s);
    }
    const int fl = a.flags & ~FFM_EPI_RANKOP;
    // BatchNorm-backward column sums exist in the kernels instantiated with the bit only (cases below; rank <= 16)
    if ((fl & FFM_EPI_BNBWD) && !(rk && (fl & ~FFM_EPI_RESIDUAL) == (FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_BNBWD)) && fl != FFM_EPI_BNBWD) return FFM_EUNSUP;
    if ((fl & FFM_EPI_BNBWD) && (fl & FFM_EPI_LORA) && a.rank > 16) return FFM_EUNSUP;
    if ((fl & FFM_EPI_LORA) && a.rank > 16) {                             // rank-r update on the VALU: generic-flag kernels
        if (rk) return dtype == FFM_BF16 ? launch_gemm<bf16_t, true, -1, false, true>(a, s) : launch_gemm<float, true, -1, false, true>(a, s);
        return dtype == FFM_BF16 ? launch_gemm<bf16_t, false, -1, false, true>(a, s) : launch_gemm<float, false, -1, false, true>(a, s);
    }
    // fewer tiles than CUs and a K loop worth pipelining (16-bit operands): the four-stage ring, one block per CU
    // (FFM_GEMM_DEEP=0: A/B runs)
    static const bool deep_on = !(getenv("FFM_GEMM_DEEP") && getenv("FFM_GEMM_DEEP")[0] == '0');
    const long tiles_ = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const bool deep = deep_on && dtype == FFM_BF16 && tiles_ <= 256 && (size_t)a.K * es / KT_BYTES >= 6;
#define FFM_GEMM_CASE(RKB, F)                                                                   \
    case F:                                                                                      \
        if (deep) {                          /* (its LDS need exceeds 160 KB for this rank / epilogue: two-buffer loop) */ \
            const int e4 = launch_gemm<bf16_t, RKB, F, false, false, 4>(a, s);                   \
            if (e4 != FFM_EUNSUP) return e4;                                                     \
        }                                                                                        \
        return dtype == FFM_BF16 ? launch_gemm<bf16_t, RKB, F>(a, s) : launch_gemm<float, RKB, F>(a, s);
    if (rk) {
        switch (fl) {
            FFM_GEMM_CASE(true, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU)                        // c_fc forward
            FFM_GEMM_CASE(true, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL)                    // c_proj forward
            FFM_GEMM_CASE(true, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU)                    // dX of c_proj
            FFM_GEMM_CASE(true, FFM_EPI_LORA | FFM_EPI_LORA_KR)                                    // dX of c_fc
            FFM_GEMM_CASE(true, FFM_EPI_LORA)                                                      // RN50 conv1 / conv3 (no bias)
            FFM_GEMM_CASE(true, FFM_EPI_BIAS | FFM_EPI_LORA)                                       // RN50 attention-pool projections
            FFM_GEMM_CASE(true, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_RESIDUAL)                 // RN50 dX of conv1 + identity path
            FFM_GEMM_CASE(true, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_BNBWD)                    // RN50 dX of conv3 + bn2's backward sums
            FFM_GEMM_CASE(true, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_RESIDUAL | FFM_EPI_BNBWD) // dX of conv1 + identity + the NEXT block's bn3 sums
            default: return dtype == FFM_BF16 ? launch_gemm<bf16_t, true, -1>(a, s) : launch_gemm<float, true, -1>(a, s);
        }
    }
    switch (fl) {
        FFM_GEMM_CASE(false, 0)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS | FFM_EPI_RESIDUAL)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS | FFM_EPI_GELU)
        FFM_GEMM_CASE(false, FFM_EPI_DGELU)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS | FFM_EPI_LORA)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU)
        FFM_GEMM_CASE(false, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL)
        FFM_GEMM_CASE(false, FFM_EPI_LORA | FFM_EPI_LORA_KR)
        FFM_GEMM_CASE(false, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU)
        FFM_GEMM_CASE(false, FFM_EPI_BNBWD)                                                      // a plain dX product + the sums
        default: return dtype == FFM_BF16 ? launch_gemm<bf16_t, false, -1>(a, s) : launch_gemm<float, false, -1>(a, s);
    }
#undef FFM_GEMM_CASE
}

// y = sum_s part[s]  (split-K partial tiles of the implicit-GEMM convolution), 4 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, T* __restrict__ y, size_t total4,
                                                            size_t stride, int S) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 a = *reinterpret_cast<const f32x4*>(part + i * 4);
        for (int s = 1; s < S; ++s) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(part + (size_t)s * stride + i * 4);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        Vec4<T>::store(y + i * 4, a);
    }
}

// The same sum with the column sums of the STORED rows beside it (round 5): a convolution split over K has no epilogue to
// leave the BatchNorm statistics in, so the BatchNorm behind it used to read the tensor once more (RN50 layer3 / layer4:
// 7 forward and 9 backward column-sum launches per step).  A block owns R rows x 64 columns (16 column quads along the
// threads x 16 row lanes, combined through LDS in a fixed order) and writes its 64 columns of partial row
// cs_part[block][2][N]: {sum, sum of squares} (forward statistics) or, BNB, {sum g, sum g xhat} with g = value x
// (bn_mask > 0), xhat = (bn_x - mean) rstd (ffm_bn_bwd's sums, FFM_EPI_BNBWD).
inline int splitk_cs_rows(int M) { return M >= 4096 ? 32 : 8; }                 // rows per block
inline bool splitk_cs_ok(int M, int N) { return N % 4 == 0 && (M + splitk_cs_rows(M) - 1) / splitk_cs_rows(M) <= 4096; }
template <typename T, bool BNB>
__global__ __launch_bounds__(256) void splitk_reduce_cs_kernel(const float* __restrict__ part, T* __restrict__ y, int M, int N,
                                                               size_t stride, int S, int R, float* __restrict__ cs_part,
                                                               const T* __restrict__ bn_x, const T* __restrict__ bn_mask,
                                                               const float* __restrict__ bn_mean, const float* __restrict__ bn_rstd) {
    // grid (row blocks of R rows, groups of 64 columns): 16 column quads x 16 row lanes per block - the first version (one
    // block per R rows x all columns: 196 blocks at layer3) streamed the 51 MB of partial tiles at a third of the plain
    // reduction's rate
    __shared__ float red[2][256][4];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64 + q * 4;
    const bool cok = c0 < N;
    const int r0 = blockIdx.x * R, r1 = (r0 + R) < M ? (r0 + R) : M;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = cs, mu = cs, rs = cs;
    if (BNB && cok) {
        mu = *reinterpret_cast<const f32x4*>(bn_mean + c0);
        rs = *reinterpret_cast<const f32x4*>(bn_rstd + c0);
    }
    if (cok) {
        for (int r = r0 + rl; r < r1; r += 16) {
            const size_t o = (size_t)r * N + c0;
            f32x4 a = *reinterpret_cast<const f32x4*>(part + o);
#pragma unroll 8
            for (int s = 1; s < S; ++s) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(part + (size_t)s * stride + o);
                a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
            }
            Vec4<T>::store(y + o, a);
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, mv = {1.f, 1.f, 1.f, 1.f};
            if (BNB) {
                xv = Vec4<T>::load(bn_x + o);
                if (bn_mask) mv = Vec4<T>::load(bn_mask + o);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float st = Elem<T>::to_f(Elem<T>::from_f(a[e]));
                if (BNB) {
                    if (!(mv[e] > 0.f)) st = 0.f;
                    cs[e] += st;
                    cq[e] += st * (xv[e] - mu[e]) * rs[e];
                } else {
                    cs[e] += st;
                    cq[e] += st * st;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = cs[e]; red[1][threadIdx.x][e] = cq[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int j = threadIdx.x >> 6, cl = threadIdx.x & 63, c = blockIdx.y * 64 + cl;
        if (c < N) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < 16; ++l) t += red[j][l * 16 + (cl >> 2)][cl & 3];
            cs_part[((size_t)blockIdx.x * 2 + j) * N + c] = t;
        }
    }
}

namespace {
// ---- 3x3 convolutions with FEW output channels (N = 32 / 64: RN50's stem and layer1), where a 128-wide tile spends
// half or three quarters of its MFMAs on padding: block tile 128 x NB, wave w owns rows [32 w, 32 w + 32) x all NB columns,
// A through the implicit-im2col loader above, the whole weight K-tile (NB rows x 128 B) beside it, two LDS buffers,
// three blocks per CU.  Epilogue: plain store through LDS (8-element row segments) + optional column sums of the STORED
// values (colstat_part, as the 128x128 kernel writes them) taken straight from the accumulators.
template <typename T, int NB, bool BNB = false>
__global__ __launch_bounds__(256, 3) void conv_narrow_kernel(gemm_kargs px) {
    const ffm_gemm_args& p = px.g;
    typedef typename Mma16<T>::frag_t frag_t;
    constexpr int NF = NB / 16;
    constexpr int WT_BYTES = NB * KT_BYTES;           // weight K-tile
    constexpr int BUFN = TILE_BYTES + WT_BYTES;
    constexpr int CLD = NB + 4;                       // padded f32 row of the C stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = tm * BM, n0 = blockIdx.y * NB;     // N % NB == 0 (checked by the launcher)
    const int nk = (int)((size_t)p.K * sizeof(T) / KT_BYTES);
    const int rsub = lane >> 3, slot = lane & 7;
    const char* Wb = reinterpret_cast<const char*>(p.b) + (size_t)n0 * (size_t)p.ldb * sizeof(T);

    auto stage_w = [&](int kbyte0, char* dst) {       // NB / 8 wave-instructions of 1 KiB, NB / 32 per wave
#pragma unroll
        for (int q = 0; q < NB / 32; ++q) {
            const int inst = wave * (NB / 32) + q;
            const int row = inst * 8 + rsub;
            const char* src = Wb + (size_t)row * (size_t)p.ldb * sizeof(T) + kbyte0 + ((slot ^ rsub) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[2][NF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    ConvA<T> cva;
    cva.init(px, m0, wave, lane, 0);
    cva.stage(px, smem, wave);
    stage_w(0, smem + TILE_BYTES);
    __syncthreads();

    const int frow = lane & 15, fgrp = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const char* As = smem + cur * BUFN;
        const char* Bs = As + TILE_BYTES;
        if (kt + 1 < nk) {
            char* An = smem + (cur ^ 1) * BUFN;
            cva.stage(px, An, wave);
            stage_w((kt + 1) * KT_BYTES, An + TILE_BYTES);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + fgrp;
            frag_t af[2], bf[NF];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wave * 32 + i * 16 + frow;
                af[i] = *reinterpret_cast<const frag_t*>(As + ra * KT_BYTES + ((chunk ^ (ra & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int rb = j * 16 + frow;
                bf[j] = *reinterpret_cast<const frag_t*>(Bs + rb * KT_BYTES + ((chunk ^ (rb & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) Mma16<T>::mma(acc[i][j], af[i], bf[j]);
        }
        __syncthreads();   // next tile landed (the compiler drains vmcnt before the barrier); cur is free
    }

    // ---- epilogue: accumulators -> LDS [128][CLD] f32 -> 8-element row segments
    float* Cs = reinterpret_cast<float*>(smem);
    float* Red = Cs + BM * CLD;                       // column sums of the four waves [2][4][NB]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                Cs[(wave * 32 + i * 16 + fgrp * 4 + e) * CLD + j * 16 + frow] = acc[i][j][e];
    if (!BNB && p.colstat_part) {
        // lane (frow, fgrp) holds column j*16 + frow of rows fgrp*4 + e: its 8 rows, then the 4 row groups of the wave
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m0 + wave * 32 + i * 16 + fgrp * 4 + e;
                    const float st = row < p.M ? Elem<T>::to_f(Elem<T>::from_f(acc[i][j][e])) : 0.f;
                    s0 += st;
                    s1 += st * st;
                }
            s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
            s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
            if (fgrp == 0) {
                Red[wave * NB + j * 16 + frow] = s0;
                Red[(4 + wave) * NB + j * 16 + frow] = s1;
            }
        }
    }
    __syncthreads();
    {
        constexpr int VN = 8;                         // elements per row segment (16 B of bf16, 32 B of f32)
        constexpr int CPR = NB / VN;                  // segments per row
        constexpr int RPP = 256 / CPR;                // rows per pass
        T* C = reinterpret_cast<T*>(p.c);
        const int cseg = tid % CPR, r0 = tid / CPR;
        // FFM_EPI_BNBWD (BNB instantiations): {sum g, sum g xhat} of the stored rows, g = value x (ReLU output > 0), from the
        // row segments as they leave (the BatchNorm's input / output rows are read as 16-byte segments beside them)
        float cs[VN], cq[VN], bmu[VN], brs[VN];
#pragma unroll
        for (int c = 0; c < VN; ++c) cs[c] = cq[c] = bmu[c] = brs[c] = 0.f;
        if constexpr (BNB) {
#pragma unroll
            for (int c = 0; c < VN; ++c) { bmu[c] = p.bn_mean[n0 + cseg * VN + c]; brs[c] = p.bn_rstd[n0 + cseg * VN + c]; }
        }
#pragma unroll
        for (int ps = 0; ps < BM / RPP; ++ps) {
            const int row = ps * RPP + r0;
            if (m0 + row < p.M) {
                float v[VN];
#pragma unroll
                for (int c = 0; c < VN; ++c) v[c] = Cs[row * CLD + cseg * VN + c];
                const size_t off = (size_t)(m0 + row) * p.ldc + n0 + cseg * VN;
                Vec8<T>::store(C + off, v);
                if constexpr (BNB) {
                    float xv[VN], mv[VN];
                    Vec8<T>::load(reinterpret_cast<const T*>(p.bn_x) + off, xv);
                    if (p.bn_mask) Vec8<T>::load(reinterpret_cast<const T*>(p.bn_mask) + off, mv);
#pragma unroll
                    for (int c = 0; c < VN; ++c) {
                        float g = Elem<T>::to_f(Elem<T>::from_f(v[c]));
                        if (p.bn_mask && !(mv[c] > 0.f)) g = 0.f;
                        cs[c] += g;
                        cq[c] += g * (xv[c] - bmu[c]) * brs[c];
                    }
                }
            }
        }
        if constexpr (BNB) {
            // the RPP row lanes of a column segment meet in LDS (the C stage is read by now: barrier first) and are added
            // in a fixed order
            __syncthreads();
            float* R0 = Cs;                           // [RPP][NB] x 2
            float* R1 = Cs + RPP * NB;
#pragma unroll
            for (int c = 0; c < VN; ++c) { R0[r0 * NB + cseg * VN + c] = cs[c]; R1[r0 * NB + cseg * VN + c] = cq[c]; }
            __syncthreads();
            if (tid < NB) {
                float s0 = 0.f, s1 = 0.f;
                for (int l = 0; l < RPP; ++l) { s0 += R0[l * NB + tid]; s1 += R1[l * NB + tid]; }
                p.colstat_part[((size_t)tm * 2) * p.N + n0 + tid] = s0;
                p.colstat_part[((size_t)tm * 2 + 1) * p.N + n0 + tid] = s1;
            }
        }
    }
    if (!BNB && p.colstat_part && tid < NB) {
        const float s0 = (Red[tid] + Red[NB + tid]) + (Red[2 * NB + tid] + Red[3 * NB + tid]);
        const float s1 = (Red[4 * NB + tid] + Red[5 * NB + tid]) + (Red[6 * NB + tid] + Red[7 * NB + tid]);
        p.colstat_part[((size_t)tm * 2) * p.N + n0 + tid] = s0;
        p.colstat_part[((size_t)tm * 2 + 1) * p.N + n0 + tid] = s1;
    }
}

template <typename T, int NB, bool BNB = false>
int launch_conv_narrow(const gemm_kargs& ka, hipStream_t s) {
    constexpr int ring = 2 * (TILE_BYTES + NB * KT_BYTES), epi = BM * (NB + 4) * 4 + 8 * NB * 4;
    const int tiles = (ka.g.M + BM - 1) / BM;
    hipLaunchKernelGGL((conv_narrow_kernel<T, NB, BNB>), dim3(tiles, ka.g.N / NB), dim3(256), ring > epi ? ring : epi, s, ka);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// K slices ffm_conv3x3_nhwc will use (1: one launch with the full epilogue)
int conv_ksplit_2buf(int M, int N, int Kp, size_t es, bool scratch, int64_t scratch_elems) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk = (int)((size_t)Kp * es / KT_BYTES);
    if (!(scratch && tiles < 160 && nk >= 16 && N % 4 == 0)) return 1;
    int S = 512 / tiles;
    if (S > 8) S = 8;
    if (S > nk / 4) S = nk / 4;
    while (S > 1 && (int64_t)S * M * N > scratch_elems) --S;
    if (S <= 1) return 1;
    const int per = (nk + S - 1) / S;
    return (nk + per - 1) / per;                                    // no empty slice
}
// Round 5: with the four-stage ring a block that has its CU to itself no longer waits out a DMA round trip per K step, so
// a convolution of few tiles wants at most 256 blocks (one per CU) instead of 512 on two per CU: the split shrinks to what
// keeps tiles x S <= 256 (layer3: 98 tiles, 3 -> 2 slices; layer4: 52 tiles, 6 -> 4) and the launch takes the deep ring.
// deep (out): this launch runs the four-stage ring.  FFM_CONV_DEEP=0: the two-buffer plan (A/B runs).
int conv_ksplit(int M, int N, int Kp, size_t es, bool scratch, int64_t scratch_elems, bool* deep = nullptr) {
    static const bool deep_on = !(getenv("FFM_CONV_DEEP") && getenv("FFM_CONV_DEEP")[0] == '0');
    int S = conv_ksplit_2buf(M, N, Kp, es, scratch, scratch_elems);
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk = (int)((size_t)Kp * es / KT_BYTES);
    bool d = false;
    if (deep_on && es == 2 && tiles <= 256 && N != 32 && N != 64) {
        int Sd = S;
        while (Sd > 1 && tiles * Sd > 256) --Sd;
        if (Sd > 1) {
            const int per = (nk + Sd - 1) / Sd;
            Sd = (nk + per - 1) / per;
        }
        if (nk / Sd >= 6) { S = Sd; d = true; }
    }
    if (deep) *deep = d;
    return S;
}
}  // namespace

extern "C" int ffm_conv3x3_colstat_rows(int B, int H, int W, int C, int N, int Kp, int64_t scratch_elems, int dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0 || (dtype != FFM_BF16 && dtype != FFM_F32)) return FFM_EINVAL;
    const int M = B * H * W;
    if (conv_ksplit(M, N, Kp, dtype == FFM_BF16 ? 2 : 4, scratch_elems > 0, scratch_elems) > 1)   // the split-K sum leaves them
        return splitk_cs_ok(M, N) ? (M + splitk_cs_rows(M) - 1) / splitk_cs_rows(M) : 0;
    return (M + BM - 1) / BM;
}

static int conv3x3_impl(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp, const void* zeros,
                        float* splitk_scratch, int64_t scratch_elems, float* colstat_part, const void* bn_x,
                        const void* bn_mask, const float* bn_mean, const float* bn_rstd, int dtype, void* stream);

extern "C" int ffm_conv3x3_nhwc(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp,
                                const void* zeros, float* splitk_scratch, int64_t scratch_elems, float* colstat_part,
                                int dtype, void* stream) {
    return conv3x3_impl(x, w, y, B, H, W, C, N, Kp, zeros, splitk_scratch, scratch_elems, colstat_part, nullptr, nullptr, nullptr,
                        nullptr, dtype, stream);
}

extern "C" int ffm_conv3x3_nhwc_bnbwd(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp,
                                      const void* zeros, float* splitk_scratch, int64_t scratch_elems, float* colstat_part,
                                      const void* bn_x, const void* bn_mask, const float* bn_mean, const float* bn_rstd,
                                      int dtype, void* stream) {
    if (!colstat_part || !bn_x || !bn_mean || !bn_rstd || (((uintptr_t)bn_x | (uintptr_t)bn_mask) & 15)) return FFM_EINVAL;
    // (ask ffm_conv3x3_colstat_rows first: 0 = this launch cannot leave them)
    if (conv_ksplit(B * H * W, N, Kp, dtype == FFM_BF16 ? 2 : 4, splitk_scratch != nullptr, scratch_elems) > 1 &&
        !splitk_cs_ok(B * H * W, N))
        return FFM_EUNSUP;
    return conv3x3_impl(x, w, y, B, H, W, C, N, Kp, zeros, splitk_scratch, scratch_elems, colstat_part, bn_x, bn_mask, bn_mean,
                        bn_rstd, dtype, stream);
}

static int conv3x3_impl(const void* x, const void* w, void* y, int B, int H, int W, int C, int N, int Kp, const void* zeros,
                        float* splitk_scratch, int64_t scratch_elems, float* colstat_part, const void* bn_x,
                        const void* bn_mask, const float* bn_mean, const float* bn_rstd, int dtype, void* stream) {
    if (!x || !w || !y || !zeros || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return FFM_EINVAL;
    if (dtype != FFM_BF16 && dtype != FFM_F32) return FFM_EINVAL;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    if (((size_t)C * es) % 16 || ((size_t)Kp * es) % KT_BYTES || Kp < 9 * C || N % 8) return FFM_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)zeros) & 15) return FFM_EINVAL;
    if ((long long)B * H * W > 0x7fffffffLL) return FFM_EUNSUP;
    gemm_kargs ka;
    ffm_gemm_args& a = ka.g;
    a = ffm_gemm_args{};
    a.a = x; a.b = w; a.c = y;
    a.M = B * H * W; a.N = N; a.K = Kp;
    a.lda = C; a.ldb = Kp; a.ldc = N;
    ka.conv_h = H; ka.conv_w = W; ka.conv_c = C; ka.conv_zero = zeros;
    ka.ksplit = 1; ka.part = nullptr;
    hipStream_t s = (hipStream_t)stream;
    // Few output tiles and a long K (layer3 / layer4: 14 x 14 and 7 x 7 maps, K = 2304 / 4608): split K over grid.y so
    // that the launch fills the chip; the fp32 partial tiles are summed by one more small kernel.
    bool deep = false;
    const int S = conv_ksplit(a.M, a.N, Kp, es, splitk_scratch != nullptr, scratch_elems, &deep);
    if (S > 1) { ka.ksplit = S; ka.part = splitk_scratch; }
    a.colstat_part = S > 1 ? nullptr : colstat_part;               // (split over K: the sums leave with the reduction below)
    const bool bnb = bn_x != nullptr;                              // FFM_EPI_BNBWD (split over K: in the reduction below)
    if (bnb) { a.bn_x = bn_x; a.bn_mask = bn_mask; a.bn_mean = bn_mean; a.bn_rstd = bn_rstd; }
    if (bnb && S == 1) a.flags = FFM_EPI_BNBWD;                    // (the split launches run the plain kernels)
    // The stem and layer1 (N = 32 / 64): 128 x N tiles, three blocks per CU (81 -> 33 / 43 us at 112 x 112, 40 -> 25 us at
    // 56 x 56).  FFM_CONV_NARROW=off: the 128x128 kernel (A/B runs); FFM_CONV_NARROW=<t>: also N = 128 / 256 / ... as 64-wide
    // column tiles when the launch has fewer than t 128x128 tiles (measured at t = 512 / 1000 on RN50 bs 32: no gain)
    const int t128 = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static const char* nenv = getenv("FFM_CONV_NARROW");
    static const int narrow_max = nenv ? (nenv[0] == 'o' ? -1 : atoi(nenv)) : 0;
    if (S == 1 && narrow_max >= 0 && (a.N == 32 || a.N == 64 || (a.N % 64 == 0 && t128 < narrow_max))) {
        if (bnb) {
            if (a.N == 32) return dtype == FFM_BF16 ? launch_conv_narrow<bf16_t, 32, true>(ka, s) : launch_conv_narrow<float, 32, true>(ka, s);
            return dtype == FFM_BF16 ? launch_conv_narrow<bf16_t, 64, true>(ka, s) : launch_conv_narrow<float, 64, true>(ka, s);
        }
        if (a.N == 32) return dtype == FFM_BF16 ? launch_conv_narrow<bf16_t, 32>(ka, s) : launch_conv_narrow<float, 32>(ka, s);
        return dtype == FFM_BF16 ? launch_conv_narrow<bf16_t, 64>(ka, s) : launch_conv_narrow<float, 64>(ka, s);
    }
    if (bnb && S == 1) {
        if (deep && dtype == FFM_BF16) return launch_gemm<bf16_t, false, FFM_EPI_BNBWD, true, false, 4>(a, s, &ka);
        return dtype == FFM_BF16 ? launch_gemm<bf16_t, false, FFM_EPI_BNBWD, true>(a, s, &ka) : launch_gemm<float, false, FFM_EPI_BNBWD, true>(a, s, &ka);
    }
    const int e = (deep && dtype == FFM_BF16) ? launch_gemm<bf16_t, false, 0, true, false, 4>(a, s, &ka)
                  : dtype == FFM_BF16 ? launch_gemm<bf16_t, false, 0, true>(a, s, &ka) : launch_gemm<float, false, 0, true>(a, s, &ka);
    if (e || ka.ksplit <= 1) return e;
    const size_t total = (size_t)a.M * a.N, total4 = total / 4;
    if (colstat_part && splitk_cs_ok(a.M, a.N)) {
        const int R = splitk_cs_rows(a.M), nb = (a.M + R - 1) / R;
#define FFM_SKCS(T, B_) hipLaunchKernelGGL((splitk_reduce_cs_kernel<T, B_>), dim3(nb, (a.N + 63) / 64), dim3(256), 0, s, ka.part, (T*)y, a.M, a.N, total, \
                                           ka.ksplit, R, colstat_part, (const T*)bn_x, (const T*)bn_mask, bn_mean, bn_rstd)
        if (dtype == FFM_BF16) { if (bnb) FFM_SKCS(bf16_t, true); else FFM_SKCS(bf16_t, false); }
        else { if (bnb) FFM_SKCS(float, true); else FFM_SKCS(float, false); }
#undef FFM_SKCS
        FFM_CHECK_LAUNCH();
        return FFM_OK;
    }
    size_t blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, ka.part, (bf16_t*)y, total4, total, ka.ksplit);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, ka.part, (float*)y, total4, total, ka.ksplit);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_gemm_tiles_m(int M, int N, int K, int flags, int rank, int dtype, int packed) {
    const int cfg = ffm_panel_select(M, N, K, flags, rank, dtype, packed != 0);
    return cfg >= 0 ? ffm_panel_ds_rows(M, N, cfg) : (M + BM - 1) / BM;
}

extern "C" int ffm_gemm_lgrad_rows(int M, int N, int K, int flags, int rank, int dtype, int packed) {
    const int cfg = ffm_panel_select(M, N, K, flags | FFM_EPI_LGRAD, rank, dtype, packed != 0);
    return cfg >= 0 ? (M + 16 * FFM_PANEL_CFGS[cfg].mf - 1) / (16 * FFM_PANEL_CFGS[cfg].mf) : FFM_EUNSUP;
}

extern "C" int ffm_gemm_tiles_n(int M, int N, int K, int flags, int rank, int dtype, int packed) {
    const int cfg = ffm_panel_select(M, N, K, flags, rank, dtype, packed != 0);
    if (cfg >= 0) return ffm_panel_tiles_n(N, cfg);
    if (flags & (FFM_EPI_ROWSTATS | FFM_EPI_LNIN | FFM_EPI_LNB_STAT | FFM_EPI_LNB_APPLY)) return FFM_EUNSUP;
    return (N + BN - 1) / BN;
}

extern "C" int ffm_gemm_tile_shape(int M, int N, int K, int flags, int rank, int dtype, int packed, int32_t* shape3) {
    const int cfg = ffm_panel_select(M, N, K, flags, rank, dtype, packed != 0);
    if (shape3) {
        shape3[0] = cfg >= 0 ? 16 * FFM_PANEL_CFGS[cfg].mf : BM;
        shape3[1] = cfg >= 0 ? ffm_panel_bn(FFM_PANEL_CFGS[cfg]) : BN;
        shape3[2] = cfg >= 0 ? FFM_PANEL_CFGS[cfg].pw * FFM_PANEL_CFGS[cfg].per_cu : 8;      // waves per CU
    }
    return cfg;
}

extern "C" int ffm_lora_pack_multi(const ffm_pack_desc* descs_dev, int ndesc, int max_K, int dtype, void* stream) {
    if (!descs_dev || ndesc <= 0 || max_K <= 0) return FFM_EINVAL;
    dim3 grid((RK_ROWS * max_K + 255) / 256, ndesc);
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((lora_pack_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, descs_dev);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((lora_pack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, descs_dev);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_lora_pack_ln(const ffm_pack_desc* descs_dev, int ndesc, int dtype, void* stream) {
    if (!descs_dev || ndesc <= 0) return FFM_EINVAL;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((lora_pack_ln_kernel<bf16_t>), dim3(ndesc), dim3(256), 0, (hipStream_t)stream, descs_dev);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((lora_pack_ln_kernel<float>), dim3(ndesc), dim3(256), 0, (hipStream_t)stream, descs_dev);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
