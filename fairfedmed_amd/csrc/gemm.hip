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
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, KT_BYTES = 128;
constexpr int TILE_BYTES = BM * KT_BYTES;            // 16 KiB per operand per buffer
constexpr int MAIN_LDS = 4 * TILE_BYTES;             // A0 B0 A1 B1
constexpr int CS_LD = 132;                           // padded f32 row of the C stage
constexpr int CS_ROWS = 64;                          // epilogue runs in two 64-row halves

__host__ __device__ constexpr int epi_lds_bytes(int r) {
    return CS_ROWS * CS_LD * 4 + r * BN * 4 + CS_ROWS * r * 4;
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

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(ffm_gemm_args p) {
    typedef typename Mma16<T>::frag_t frag_t;
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
    const int nk = (int)((size_t)p.K * sizeof(T) / KT_BYTES);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: tile 0 -> buffer 0
    stage_tile<T>(A, p.lda, m0, p.M, 0, smem, wave, lane);
    stage_tile<T>(B, p.ldb, n0, p.N, 0, smem + TILE_BYTES, wave, lane);
    __syncthreads();

    const int frow = lane & 15, fgrp = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        char* As = smem + cur * 2 * TILE_BYTES;
        char* Bs = As + TILE_BYTES;
        if (kt + 1 < nk) {
            char* An = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile<T>(A, p.lda, m0, p.M, (kt + 1) * KT_BYTES, An, wave, lane);
            stage_tile<T>(B, p.ldb, n0, p.N, (kt + 1) * KT_BYTES, An + TILE_BYTES, wave, lane);
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
        }
        __syncthreads();   // next tile landed (compiler drains vmcnt before the barrier); cur is free
    }

    // ---------------- epilogue: two halves of 64 rows through LDS ----------
    float* Cs = reinterpret_cast<float*>(smem);
    const bool has_lora = (p.flags & FFM_EPI_LORA) != 0;
    const int r = has_lora ? p.rank : 0;
    float* Ls = Cs + CS_ROWS * CS_LD;                 // LoRA matrix tile [r][BN]
    float* Ts = Ls + r * BN;                          // ts rows [64][r]
    T* C = reinterpret_cast<T*>(p.c);

    if (has_lora) {
        // LoRA matrix tile: Ls[j][n] for n0..n0+127
        for (int idx = tid; idx < r * BN; idx += 256) {
            const int j = idx / BN, n = idx % BN;
            float v = 0.f;
            if (n0 + n < p.N)
                v = (p.flags & FFM_EPI_LORA_KR) ? p.lw[(size_t)(n0 + n) * r + j] : p.lw[(size_t)j * p.N + n0 + n];
            Ls[idx] = v;
        }
    }

    const int ecol = (tid & 15) * 8;                  // this thread's 8 columns
    const int erow0 = tid >> 4;                       // rows erow0 + 16*i
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        Cs[(i * 16 + fgrp * 4 + e) * CS_LD + wn * 64 + j * 16 + frow] = acc[i][j][e];
        }
        if (has_lora) {
            for (int idx = tid; idx < CS_ROWS * r; idx += 256) {
                const int row = idx / r, j = idx % r;
                const int gm = m0 + half * 64 + row;
                Ts[idx] = gm < p.M ? p.ts[(size_t)gm * r + j] : 0.f;
            }
        }
        __syncthreads();
        const int gn = n0 + ecol;
        if (gn < p.N) {
            float v[4][8];
            {
                float bias8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) bias8[c] = (p.flags & FFM_EPI_BIAS) ? p.bias[gn + c] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int lrow = erow0 + 16 * i;
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(&Cs[lrow * CS_LD + ecol]);
                    const f32x4 c1 = *reinterpret_cast<const f32x4*>(&Cs[lrow * CS_LD + ecol + 4]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { v[i][c] = c0[c] + bias8[c]; v[i][4 + c] = c1[c] + bias8[4 + c]; }
                }
            }
            if (has_lora) {
                // rank-r update: this thread's 8 columns of the LoRA matrix stay in registers
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
                            const float tj = (j0 + jj) < r ? Ts[lrow * r + j0 + jj] : 0.f;
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
                if (p.flags & FFM_EPI_RESIDUAL) {
                    float rr[8];
                    Vec8<T>::load(reinterpret_cast<const T*>(p.res) + off, rr);
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[i][c] += rr[c];
                }
                if (p.flags & FFM_EPI_DGELU) {
                    float pre[8];
                    Vec8<T>::load(reinterpret_cast<const T*>(p.aux) + off, pre);
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[i][c] *= Act<T>::gelu_grad(pre[c]);
                }
                Vec8<T>::store(C + off, v[i]);
                if (p.flags & FFM_EPI_GELU) {
                    float a[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) a[c] = Act<T>::gelu(Elem<T>::to_f(Elem<T>::from_f(v[i][c])));
                    Vec8<T>::store(reinterpret_cast<T*>(p.c2) + off, a);
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
int launch_gemm(const ffm_gemm_args& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int r = (a.flags & FFM_EPI_LORA) ? a.rank : 0;
    int lds = epi_lds_bytes(r);
    if (lds < MAIN_LDS) lds = MAIN_LDS;
    hipLaunchKernelGGL((gemm_nt_kernel<T>), dim3(tiles), dim3(256), lds, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

}  // namespace

extern "C" int ffm_gemm_nt(const ffm_gemm_args* args, int dtype, void* stream) {
    if (!args || !args->a || !args->b || !args->c) return FFM_EINVAL;
    const ffm_gemm_args& a = *args;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    if (dtype != FFM_BF16 && dtype != FFM_F32) return FFM_EINVAL;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return FFM_EINVAL;
    if (((size_t)a.K * es) % KT_BYTES != 0 || a.N % 8 != 0) return FFM_EINVAL;
    if (((size_t)a.lda * es) % 16 || ((size_t)a.ldb * es) % 16 || ((size_t)a.ldc * es) % 16) return FFM_EINVAL;
    if (((uintptr_t)a.a | (uintptr_t)a.b | (uintptr_t)a.c) & 15) return FFM_EINVAL;
    if (a.lda < a.K || a.ldb < a.K || a.ldc < a.N) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_LORA) && (a.rank <= 0 || a.rank > FFM_MAX_RANK || !a.ts || !a.lw)) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_BIAS) && !a.bias) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_RESIDUAL) && (!a.res || ((uintptr_t)a.res & 15))) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_GELU) && (!a.c2 || ((uintptr_t)a.c2 & 15))) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_DGELU) && (!a.aux || ((uintptr_t)a.aux & 15))) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    return dtype == FFM_BF16 ? launch_gemm<bf16_t>(a, s) : launch_gemm<float>(a, s);
}
