// Rank-r FairLoRA kernels (HBM-bound; VALU + wave shuffles + LDS, no MFMA):
//   lora_down         t = x P, ts = scaling * t * s_b, dS partials
//   lora_grad_partial part[s] = sum_{rows of split s} x^T v   (dA / dB without forming dW)
//   reduce_partials   deterministic second stage
//
// Both streaming kernels use "column-owner" lanes: a lane owns 16 bytes of every
// row it visits (8 bf16 / 4 f32 columns), so each wave-instruction reads 1 KiB
// of one row, fully coalesced, and the rank-r operand of that lane lives in
// registers for the whole kernel.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// lora_down.  Block = LD_WAVES waves; the rank-r operand P is staged ONCE per
// block in LDS (in the activation dtype, laid out so that the 64 lanes of a
// wave read consecutive 16-byte units).  Each wave owns RSUB = 64/RP rows and
// sweeps the whole K: per 16-byte chunk it loads the RSUB rows, reads that
// chunk's P slice from LDS once and reuses it for all RSUB rows.  The
// RSUB x RP partial sums are combined across the 64 lanes by a butterfly
// reduce-scatter (63 shuffles for 64 values) that leaves value (row, j) on lane
// row*RP + j, which then writes t / ts directly.
// ---------------------------------------------------------------------------
constexpr int LD_WAVES = 8;
constexpr int LD_LDS_BUDGET = 64 * 1024;

template <int V>
__device__ __forceinline__ void reduce_scatter64(float (&vals)[V], int lane) {
    // V == 64: afterwards vals[0] on lane l is the sum over all lanes of vals[l].
#pragma unroll
    for (int n = V, o = 32; o >= 1; n >>= 1, o >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float a = vals[i], b = vals[i + n / 2];
            const float keep = up ? b : a, send = up ? a : b;
            vals[i] = keep + __shfl_xor(send, o, 64);
        }
    }
}

template <typename T, int RP>
__global__ __launch_bounds__(LD_WAVES * 64) void lora_down_kernel(
    const T* __restrict__ x, int ldx, const float* __restrict__ P, int layout_rk, const float* __restrict__ S,
    const int32_t* __restrict__ attr, int M, int K, int r, int G, int rows_per_sample, float scaling,
    float lambda_group, float* __restrict__ t_out, float* __restrict__ ts_out, const float* __restrict__ t_fwd,
    float* __restrict__ ds_part, int ktile, int rs, int j0, int kq_n) {
    // r = columns handled by this pass (<= RP), starting at column j0 of arrays whose rank stride is rs.
    // The block's 8 waves are (8/kq_n) row groups x kq_n slices of K; slices are summed through LDS.
    typedef typename Elem<T>::chunk_t chunk_t;
    constexpr int CE = Elem<T>::kPerChunk;           // columns per 16-byte chunk of x
    constexpr int NQ = RP / CE;                       // 16-byte units of P per column (RP values of type T)
    constexpr int RSUB = 64 / RP;                     // rows per row group
    constexpr int ROWS = LD_WAVES * RSUB;             // max rows per block (kq_n = 1)
    static_assert(RP % CE == 0 && 64 % RP == 0, "rank padding must be a multiple of the chunk width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    chunk_t* Ps = reinterpret_cast<chunk_t*>(smem);   // unit index ((e*NQ + q) * nch + ch): P[ch*CE+e][q*CE .. +CE)
    __shared__ float Vs[ROWS * RP];
    __shared__ float Red[LD_WAVES * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave % kq_n, rg = wave / kq_n;
    const int rows_blk = (LD_WAVES / kq_n) * RSUB;
    const int row0 = blockIdx.x * rows_blk + rg * RSUB;
    float vals[RSUB * RP];
#pragma unroll
    for (int i = 0; i < RSUB * RP; ++i) vals[i] = 0.f;

    for (int k0 = 0; k0 < K; k0 += ktile) {
        const int kt = (K - k0) < ktile ? (K - k0) : ktile;
        const int nch = kt / CE;                      // chunks in this tile
        __syncthreads();
        // stage P[k0 .. k0+kt) as T.  A thread converts 4 consecutive columns at a time from 16-byte
        // loads that are all in flight together (the tile is read once per block, so its latency is
        // exposed: keep it to one round trip).
        const bool vec_ok = ((rs & 3) == 0) && ((j0 & 3) == 0) && ((K & 3) == 0) && ((k0 & 3) == 0);
#pragma unroll 2
        for (int g4 = tid; g4 < (kt + 3) / 4; g4 += LD_WAVES * 64) {
            float pv4[4][RP];                          // [column u][rank j]
            if (vec_ok && 4 * g4 + 3 < kt) {
                if (layout_rk) {
#pragma unroll
                    for (int j = 0; j < RP; ++j) {
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        if (j < r) v = *reinterpret_cast<const f32x4*>(P + (size_t)(j0 + j) * K + k0 + 4 * g4);
#pragma unroll
                        for (int u = 0; u < 4; ++u) pv4[u][j] = v[u];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int jq = 0; jq < RP / 4; ++jq) {
                            f32x4 v = {0.f, 0.f, 0.f, 0.f};
                            if (jq * 4 < r)
                                v = *reinterpret_cast<const f32x4*>(P + (size_t)(k0 + 4 * g4 + u) * rs + j0 + jq * 4);
#pragma unroll
                            for (int c = 0; c < 4; ++c) pv4[u][jq * 4 + c] = (jq * 4 + c) < r ? v[c] : 0.f;
                        }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < RP; ++j) {
                        const int kk = 4 * g4 + u;
                        float v = 0.f;
                        if (kk < kt && j < r)
                            v = layout_rk ? P[(size_t)(j0 + j) * K + k0 + kk] : P[(size_t)(k0 + kk) * rs + j0 + j];
                        pv4[u][j] = v;
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = 4 * g4 + u;
                if (kk < kt) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        chunk_t w;
#pragma unroll
                        for (int c = 0; c < CE; ++c) w[c] = Elem<T>::from_f(pv4[u][q * CE + c]);
                        Ps[((kk % CE) * NQ + q) * nch + kk / CE] = w;
                    }
                }
            }
        }
        __syncthreads();
        const int cpw = ((nch + kq_n - 1) / kq_n + 63) / 64 * 64;      // chunks per K slice
        const int ch_end = (kq + 1) * cpw < nch ? (kq + 1) * cpw : nch;
#pragma unroll 1
        for (int ch = kq * cpw + lane; ch < ch_end; ch += 64) {
            chunk_t xv[RSUB];
#pragma unroll
            for (int i = 0; i < RSUB; ++i) {
                int row = row0 + i;
                row = row < M ? row : M - 1;
                xv[i] = *reinterpret_cast<const chunk_t*>(x + (size_t)row * ldx + k0 + ch * CE);
            }
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                float pv[RP];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const chunk_t u = Ps[(e * NQ + q) * nch + ch];
#pragma unroll
                    for (int c = 0; c < CE; ++c) pv[q * CE + c] = Elem<T>::to_f(u[c]);
                }
#pragma unroll
                for (int i = 0; i < RSUB; ++i) {
                    const float xe = Elem<T>::to_f(xv[i][e]);
#pragma unroll
                    for (int j = 0; j < RP; ++j) vals[i * RP + j] += xe * pv[j];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the per-element working set small (VGPR pressure)
            }
        }
    }
    reduce_scatter64<RSUB * RP>(vals, lane);
    float tval = vals[0];
    if (kq_n > 1) {                                   // sum the K slices of this row group
        Red[wave * 64 + lane] = tval;
        __syncthreads();
        tval = 0.f;
        for (int q = 0; q < kq_n; ++q) tval += Red[(rg * kq_n + q) * 64 + lane];
    }
    const int oj = lane % RP;
    const int orow = row0 + lane / RP;
    const bool writer = (kq == 0) && orow < M && oj < r;
    if (writer) {
        const int sample = orow / rows_per_sample;
        float sbv = 0.f;
        for (int g = 0; g < G; ++g) sbv += group_mix_w(attr, sample, g, G, lambda_group) * S[g * rs + j0 + oj];
        if (t_out) t_out[(size_t)orow * rs + j0 + oj] = tval;
        if (ts_out) ts_out[(size_t)orow * rs + j0 + oj] = scaling * tval * sbv;
    }
    if (t_fwd && ds_part) {
        if (kq == 0)
            Vs[(rg * RSUB + lane / RP) * RP + oj] = writer ? scaling * t_fwd[(size_t)orow * rs + j0 + oj] * tval : 0.f;
        __syncthreads();
        if (tid < G * r) {
            const int g = tid / r, j = tid % r;
            float s = 0.f;
            for (int rr = 0; rr < rows_blk; ++rr) {
                const int grow = blockIdx.x * rows_blk + rr;
                if (grow < M) s += group_mix_w(attr, grow / rows_per_sample, g, G, lambda_group) * Vs[rr * RP + j];
            }
            ds_part[((size_t)blockIdx.x * G + g) * rs + j0 + j] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// lora_down on the matrix cores (bf16, rank <= 16, P given as [r][K] rows: the backward direction u = g B^T).
// The VALU kernel above stages P (K x r floats: 98 KB at K = 3072) into LDS in EVERY block - as many bytes again as
// the activation rows it reads - and ran at 1.2 TB/s where it stood alone on the critical path: the last down projection
// of a step (block 0 has no dX(c_fc) GEMM to ride in), 31.7 us for 38.7 MB.  Here a block is 16 rows x 4 K slices (one
// wave each): x fragments straight from global memory (16 B per lane), P fragments converted to bf16 in registers
// (the same rounding the VALU kernel applies element by element), v_mfma_f32_16x16x32_bf16, slices summed through LDS,
// then ts = scaling * t * s_b and the block's dS partial as in the panel GEMM's rank stage.
// ---------------------------------------------------------------------------
constexpr int LDM_ROWS = 16, LDM_WAVES = 4, LDM_U = 4;       // rows per block, K slices, k32 steps of loads in flight

__global__ __launch_bounds__(LDM_WAVES * 64) void lora_down_mfma_kernel(
    const bf16_t* __restrict__ x, int ldx, const float* __restrict__ P, const float* __restrict__ S,
    const int32_t* __restrict__ attr, int M, int K, int r, int G, int rows_per_sample, float scaling, float lambda_group,
    float* __restrict__ t_out, float* __restrict__ ts_out, const float* __restrict__ t_fwd, float* __restrict__ ds_part) {
    __shared__ f32x4 red[LDM_WAVES][64];
    __shared__ float dsum[(FFM_MAX_GROUPS + 2) * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kgp = lane >> 4;                  // A: row col, B: rank slot col; 8 consecutive k each
    const int row0 = blockIdx.x * LDM_ROWS;
    int arow = row0 + col;
    arow = arow < M ? arow : M - 1;
    const int ksl = K / LDM_WAVES;                                // K slice of this wave (a multiple of 32)
    const bf16_t* xa = x + (size_t)arow * ldx + wave * ksl + kgp * 8;
    const float* pb = P + (size_t)(col < r ? col : 0) * K + wave * ksl + kgp * 8;
    const bool bok = col < r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int steps = ksl >> 5;
    for (int s0 = 0; s0 < steps; s0 += LDM_U) {
        bf16x8 av[LDM_U];
        f32x4 b0[LDM_U], b1[LDM_U];
#pragma unroll
        for (int u = 0; u < LDM_U; ++u) {
            const int st = s0 + u < steps ? s0 + u : steps - 1;  // (clamped: the surplus products are skipped below)
            av[u] = *reinterpret_cast<const bf16x8*>(xa + st * 32);
            b0[u] = *reinterpret_cast<const f32x4*>(pb + st * 32);
            b1[u] = *reinterpret_cast<const f32x4*>(pb + st * 32 + 4);
        }
#pragma unroll
        for (int u = 0; u < LDM_U; ++u) {
            if (s0 + u >= steps) break;
            bf16x8 bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bv[e] = bok ? (bf16_t)b0[u][e] : (bf16_t)0.f;
                bv[4 + e] = bok ? (bf16_t)b1[u][e] : (bf16_t)0.f;
            }
            // D[j][row] += sum_k P[j][k] x[row][k]: lane (column = row `col`... ) - operand order (P, x) puts the rank slot on
            // the accumulator's row index 4 kgp + e and the activation row on its column `col`
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av[u], acc, 0, 0, 0);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    f32x4 t4 = red[0][lane];
#pragma unroll
    for (int w = 1; w < LDM_WAVES; ++w) {
        const f32x4 o = red[w][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[e] += o[e];
    }
    // lane: activation row `col` of the block, rank slots j = 4 kgp + e
    const int grow = row0 + col;
    const bool rok = grow < M;
    const int sample = (rok ? grow : M - 1) / rows_per_sample;
    const int ga = attr ? attr[sample] : -1;
    const float w_own = lambda_group, w_oth = (1.0f - lambda_group) / (float)(G > 1 ? G - 1 : 1), w_uni = 1.0f / (float)G;
    const bool do_ds = t_fwd && ds_part;
    float wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = 4 * kgp + e;
        wv[e] = 0.f;
        if (j < r && rok) {
            float sb = 0.f;
            for (int g = 0; g < G; ++g) sb += (ga < 0 ? w_uni : (ga == g ? w_own : w_oth)) * S[g * r + j];
            if (t_out) t_out[(size_t)grow * r + j] = t4[e];
            if (ts_out) ts_out[(size_t)grow * r + j] = scaling * t4[e] * sb;
            if (do_ds) wv[e] = scaling * t_fwd[(size_t)grow * r + j] * t4[e];
        }
    }
    if (!do_ds) return;
    // dS partial of the block's 16 rows: sum over the rows (lanes of equal kgp: the low four lane bits) per rank slot, split
    // into all rows / rows without a group / rows of group g, combined as the panel GEMM's rank stage does
    float d_all[4], d_uni[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        d_all[e] = ga < 0 ? 0.f : wv[e];
        d_uni[e] = ga < 0 ? wv[e] : 0.f;
    }
    auto rows16 = [](float v) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        d_all[e] = rows16(d_all[e]);
        d_uni[e] = rows16(d_uni[e]);
    }
    if (col == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dsum[0 * 16 + 4 * kgp + e] = d_all[e];
            dsum[1 * 16 + 4 * kgp + e] = d_uni[e];
        }
    }
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float o = rows16(ga == g ? wv[e] : 0.f);
            if (col == 0) dsum[(2 + g) * 16 + 4 * kgp + e] = o;
        }
    }
    // (one wave: the LDS writes above are visible to it after the wait the compiler puts in front of the reads)
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (lane < G * r) {
        const int g = lane / r, j = lane % r;
        ds_part[((size_t)blockIdx.x * G + g) * r + j] = w_uni * dsum[16 + j] + w_oth * dsum[j] + (w_own - w_oth) * dsum[(2 + g) * 16 + j];
    }
    for (int i = lane + 64; i < G * r; i += 64) {
        const int g = i / r, j = i % r;
        ds_part[((size_t)blockIdx.x * G + g) * r + j] = w_uni * dsum[16 + j] + w_oth * dsum[j] + (w_own - w_oth) * dsum[(2 + g) * 16 + j];
    }
}

static bool down_mfma_ok(int M, int K, int r, int dtype, int layout_rk) {
    return dtype == FFM_BF16 && layout_rk && r <= 16 && M >= 1024 && K % (32 * LDM_WAVES) == 0;
}

// K slices per block: enough waves to fill the chip (>= ~2000) while every lane of a slice has work
static int down_kq(int K, int es) {
    const int nch = K * es / 16;
    return nch >= 256 ? 4 : nch >= 96 ? 2 : 1;
}
static int down_rp(int r, int dtype) { return r <= 4 && dtype == FFM_F32 ? 4 : r <= 8 ? 8 : 16; }

template <typename T, int RP>
int launch_down(const void* x, int ldx, const float* P, int layout_rk, const float* S, const int32_t* attr, int M,
                int K, int r, int G, int rps, float scaling, float lam, float* t, float* ts, const float* t_fwd,
                float* ds_part, int rs, int j0, hipStream_t s) {
    constexpr int CE = Elem<T>::kPerChunk;
    // K tile: whole K if its P image fits the LDS budget, else the largest multiple of 64 chunks that does
    int ktile = K;
    const int per_col = RP * (int)sizeof(T);
    if (K * per_col > LD_LDS_BUDGET) ktile = (LD_LDS_BUDGET / per_col) / (64 * CE) * (64 * CE);
    const int lds = ktile * per_col;
    const int kq = down_kq(K, (int)sizeof(T));
    const int rows = (LD_WAVES / kq) * (64 / RP);
    const int blocks = (M + rows - 1) / rows;
    hipLaunchKernelGGL((lora_down_kernel<T, RP>), dim3(blocks), dim3(LD_WAVES * 64), lds, s, (const T*)x, ldx, P,
                       layout_rk, S, attr, M, K, r, G, rps, scaling, lam, t, ts, t_fwd, ds_part, ktile, rs, j0, kq);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// ---------------------------------------------------------------------------
// lora_grad_partial.  Block = LG_WAVES waves over the same 64*CE columns, each
// wave sums LG_RPW rows (v[m][:] is wave-uniform -> scalar loads), then the
// waves are combined by an LDS tree; one partial per (column group, row block).
// ---------------------------------------------------------------------------
constexpr int LG_WAVES = 4;
constexpr int LG_RPW = 32;
constexpr int LG_ROWS = LG_WAVES * LG_RPW;      // rows per block (= rows per partial)

template <typename T, int RP>
__global__ __launch_bounds__(LG_WAVES * 64) void lora_grad_kernel(const T* __restrict__ x, int ldx,
                                                                   const float* __restrict__ v, int M, int K, int r,
                                                                   float* __restrict__ part, int rs, int j0) {
    constexpr int CE = Elem<T>::kPerChunk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);     // [2 waves][CE*RP][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = (blockIdx.x * 64 + lane) * CE;
    const int split = blockIdx.y;
    const bool active = k < K;
    const int m0 = split * LG_ROWS + wave * LG_RPW;
    const int m1 = (m0 + LG_RPW) < M ? (m0 + LG_RPW) : M;

    float acc[CE][RP];
#pragma unroll
    for (int e = 0; e < CE; ++e)
#pragma unroll
        for (int j = 0; j < RP; ++j) acc[e][j] = 0.f;

    const T* xc = x + (active ? k : 0);
    for (int m = m0; m < m1; m += 8) {
        typename Elem<T>::chunk_t xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = (m + u) < m1 ? (m + u) : (m1 - 1);
            xv[u] = *reinterpret_cast<const typename Elem<T>::chunk_t*>(xc + (size_t)mm * ldx);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (m + u < m1) {
                const float* vr = v + (size_t)(m + u) * rs + j0;   // wave-uniform address -> scalar loads
                float vj[RP];
#pragma unroll
                for (int j = 0; j < RP; ++j) vj[j] = (j < r) ? vr[j] : 0.f;
#pragma unroll
                for (int e = 0; e < CE; ++e) {
                    const float xe = Elem<T>::to_f(xv[u][e]);
#pragma unroll
                    for (int j = 0; j < RP; ++j) acc[e][j] += xe * vj[j];
                }
            }
        }
    }
    // tree over the 4 waves: {2,3} -> LDS -> {0,1} add; {1} -> LDS -> {0} adds
    if (wave >= 2) {
#pragma unroll
        for (int e = 0; e < CE; ++e)
#pragma unroll
            for (int j = 0; j < RP; ++j) red[((wave - 2) * CE * RP + e * RP + j) * 64 + lane] = acc[e][j];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int e = 0; e < CE; ++e)
#pragma unroll
            for (int j = 0; j < RP; ++j) acc[e][j] += red[(wave * CE * RP + e * RP + j) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int e = 0; e < CE; ++e)
#pragma unroll
            for (int j = 0; j < RP; ++j) red[(e * RP + j) * 64 + lane] = acc[e][j];
    }
    __syncthreads();
    if (wave == 0 && active) {
        float* dst = part + ((size_t)split * K + k) * rs + j0;
#pragma unroll
        for (int e = 0; e < CE; ++e)
#pragma unroll
            for (int j = 0; j < RP; ++j)
                if (j < r) dst[e * rs + j] = acc[e][j] + red[(e * RP + j) * 64 + lane];
    }
}

template <typename T, int RP>
int launch_grad(const void* x, int ldx, const float* v, int M, int K, int r, float* part, int rs, int j0,
                hipStream_t s) {
    constexpr int CE = Elem<T>::kPerChunk;
    dim3 grid((K + 64 * CE - 1) / (64 * CE), (M + LG_ROWS - 1) / LG_ROWS);
    const int lds = 2 * CE * RP * 64 * 4;
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lora_grad_kernel<T, RP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((lora_grad_kernel<T, RP>), grid, dim3(LG_WAVES * 64), lds, s, (const T*)x, ldx, v, M, K, r,
                       part, rs, j0);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// ---------------------------------------------------------------------------
// bf16 fast path of the same reduction on the matrix cores: part[split][k][j] = sum_rows x[row][k] * v[row][j]
// is the product V^T [16 x rows] . X [rows x K] with the ROW index contracted, i.e. X is needed column-major.
// One wave per block owns a 128-row x 128-column tile of X: 32 LDS-DMA pieces (4 rows x 256 B each) land it
// row-major in LDS (coalesced from HBM), ds_read_b64_tr_b16 hands it to the MFMA B operand transposed, and v goes
// into the A operand as a bf16 hi + lo pair (two MFMAs), so the result keeps fp32-level accuracy in v.
// 32 KiB of LDS per block -> 5 blocks per CU: the whole 6304 x 3072 operand (24 x 50 = 1200 tiles) is in flight at
// once and the kernel runs at the HBM rate instead of the load latency of a VALU loop.
// LDS image: 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) --
// conflict-free for the transposed reads (cdna_hip_programming.md T10, image (b)).
// ---------------------------------------------------------------------------
constexpr int LGM_ROWS = 128, LGM_COLS = 128;

__device__ __forceinline__ int lgm_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// LN: x holds RAW rows of a LayerNorm input whose normalised copy was never written (FFM_EPI_LNIN); the row factor
// rstd goes onto v, and the two rank-r row sums of the correction are formed from the v rows every block loads anyway.
struct lgm_ln {
    const float *mean, *rstd, *gamma, *beta;
};

template <int RP, bool LN>   // RP unused slots of the 16 rank slots are zero
__global__ __launch_bounds__(64) void lora_grad_mfma_kernel(const bf16_t* __restrict__ x, int ldx,
                                                            const float* __restrict__ v, int M, int K, int r,
                                                            float* __restrict__ part, int rs, int j0, lgm_ln ln) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int k0 = blockIdx.x * LGM_COLS, split = blockIdx.y, m0 = split * LGM_ROWS;

#ifdef FFM_LGM_NOLDS
    // diagnostic build (tools/step_ablate2.py): the same bytes through registers, no LDS (the result is garbage) - is it
    // the kernel's HBM traffic or its LDS footprint that slows the vision chain down?
    {
        bf16x8 sink = {};
#pragma unroll 8
        for (int piece = 0; piece < LGM_ROWS / 4; ++piece) {
            const int row = piece * 4 + (lane >> 4);
            int gm = m0 + row;
            gm = gm < M ? gm : M - 1;
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (size_t)gm * ldx + k0 + (lane & 15) * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sink[e] = (bf16_t)((float)sink[e] + (float)t[e]);
        }
        if (v[0] == 123456.f) part[lane] = (float)sink[0];
        return;
    }
#endif
    // ---- X tile -> LDS: piece = 4 rows; lane -> row 4*piece + (lane >> 4), LDS slot lane & 15
#pragma unroll
    for (int piece = 0; piece < LGM_ROWS / 4; ++piece) {
        const int row = piece * 4 + (lane >> 4);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;                              // clamped rows meet v = 0 below
        const int ch = (lane & 15) ^ lgm_swz(row);
        const bf16_t* src = x + (size_t)gm * ldx + k0 + ch * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
    }
    // ---- v -> A fragments (rank slot j = lane & 15, rows 8*(lane >> 4) .. +7 of each 32-row block), hi + lo
    const int j = lane & 15, kg = lane >> 4;
    bf16x8 ahi[4], alo[4];
    float s1 = 0.f, s2 = 0.f;                                  // LN: sum_m mean rstd v, sum_m v (this lane's rows)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int gm = m0 + rb * 32 + kg * 8 + t;
            vv[t] = (j < r && gm < M) ? v[(size_t)gm * rs + j0 + j] : 0.f;
            if constexpr (LN) {
                const int gc = gm < M ? gm : M - 1;
                const float rsd = ln.rstd[gc];
                s2 += vv[t];
                vv[t] *= rsd;
                s1 += ln.mean[gc] * vv[t];
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bf16_t h = (bf16_t)vv[t];
            ahi[rb][t] = h;
            alo[rb][t] = (bf16_t)(vv[t] - (float)h);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // single wave: orders the DMA writes before the reads

    // ---- 8 column fragments x 4 row blocks; B fragment = two transposed reads (rows 8kg..8kg+3, 8kg+4..8kg+7)
    const int q = (lane & 15) >> 2, pp = lane & 3;
    f32x4 acc[LGM_COLS / 16];
#pragma unroll
    for (int cf = 0; cf < LGM_COLS / 16; ++cf) acc[cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int R = rb * 32 + kg * 8 + q;
        const uint32_t base0 = (uint32_t)(uintptr_t)smem + 256 * R + 8 * (pp & 1);
        const uint32_t base1 = base0 + 256 * 4;
        const int s0 = lgm_swz(R), s1 = lgm_swz(R + 4);
        // all 16 transposed reads of the row block are issued before the single wait
        u32x2 lo2[LGM_COLS / 16], hi2[LGM_COLS / 16];
#pragma unroll
        for (int cf = 0; cf < LGM_COLS / 16; ++cf) {
            const int ch = cf * 2 + (pp >> 1);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo2[cf]) : "v"(base0 + 16 * (ch ^ s0)) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi2[cf]) : "v"(base1 + 16 * (ch ^ s1)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int cf = 0; cf < LGM_COLS / 16; ++cf) {
            asm volatile("" : "+v"(lo2[cf]), "+v"(hi2[cf]));
            const u32x4 packed = {lo2[cf][0], lo2[cf][1], hi2[cf][0], hi2[cf][1]};
            const bf16x8 b = __builtin_bit_cast(bf16x8, packed);
            acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[rb], b, acc[cf], 0, 0, 0);
            acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[rb], b, acc[cf], 0, 0, 0);
        }
    }
    // ---- D[jrow = 4*(lane>>4) + e][col = lane & 15] -> part[split][k0 + 16 cf + col][j0 + jrow]
    (void)RP;
    float c1[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LN) {
        // the four row groups of the wave hold disjoint rows of rank slot j: after the butterfly every lane has the sums
        // of slot (lane & 15); slot jj = 4 kg + e is fetched from lane jj
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            c1[e] = __shfl(s1, 4 * kg + e, 64);
            c2[e] = __shfl(s2, 4 * kg + e, 64);
        }
    }
#pragma unroll
    for (int cf = 0; cf < LGM_COLS / 16; ++cf) {
        const int k = k0 + cf * 16 + (lane & 15);
        float* dst = part + ((size_t)split * K + k) * rs + j0;
        float gk = 1.f, bk = 0.f;
        if constexpr (LN) { gk = ln.gamma[k]; bk = ln.beta[k]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jj = 4 * kg + e;
            if (jj < r) dst[jj] = LN ? gk * (acc[cf][e] - c1[e]) + bk * c2[e] : acc[cf][e];
        }
    }
}

int launch_grad_mfma(const void* x, int ldx, const float* v, int M, int K, int r, float* part, int rs, int j0,
                     hipStream_t s, const lgm_ln* ln = nullptr) {
    dim3 grid(K / LGM_COLS, (M + LGM_ROWS - 1) / LGM_ROWS);
#ifdef FFM_LGM_NOLDS
#define LGM_LDS 0
#else
#define LGM_LDS (LGM_ROWS * LGM_COLS * 2)
#endif
    if (ln)
        hipLaunchKernelGGL((lora_grad_mfma_kernel<16, true>), grid, dim3(64), LGM_LDS, s, (const bf16_t*)x, ldx,
                           v, M, K, r, part, rs, j0, *ln);
    else
        hipLaunchKernelGGL((lora_grad_mfma_kernel<16, false>), grid, dim3(64), LGM_LDS, s, (const bf16_t*)x, ldx,
                           v, M, K, r, part, rs, j0, lgm_ln{});
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// out[o(i)] (+)= sum_s part[s][i]; 4 split-lanes per output, combined by shuffles
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nsplit, int n,
                                                              float* __restrict__ out, int tK, int tr,
                                                              int accumulate) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (gid >> 2);                     // 4 consecutive lanes share an output
    const int q = gid & 3;
    float s = 0.f;
    if (i < n)
        for (int sp = q; sp < nsplit; sp += 4) s += part[(size_t)sp * n + i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (i >= n || q != 0) return;
    int o = i;
    if (tK > 0) {                                 // i = k * r + j  ->  j * K + k
        const int k = i / tr, j = i % tr;
        o = j * tK + k;
    }
    out[o] = accumulate ? out[o] + s : s;
}

// the same sum for MANY partial rows of a SHORT tensor (3D OCT: 100 images x 196 blocks = 19 600 partial rows of the
// 603 slice-convolution gradients, where the kernel above leaves 2 400 threads walking 4 900 rows each: 1.8 ms).
// A block owns 64 consecutive outputs; its 16 waves stride over the partial rows (256-byte coalesced reads), the wave
// sums meet in LDS and are added in wave order (deterministic).
__global__ __launch_bounds__(1024) void reduce_partials_tall_kernel(const float* __restrict__ part, int nsplit, int n,
                                                                    float* __restrict__ out, int tK, int tr, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                 // four rows in flight per wave
    if (i < n) {
        int sp = wave;
        for (; sp + 48 < nsplit; sp += 64) {
            s0 += part[(size_t)sp * n + i];
            s1 += part[(size_t)(sp + 16) * n + i];
            s2 += part[(size_t)(sp + 32) * n + i];
            s3 += part[(size_t)(sp + 48) * n + i];
        }
        for (; sp < nsplit; sp += 16) s0 += part[(size_t)sp * n + i];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave != 0 || i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][lane];
    int o = i;
    if (tK > 0) {
        const int k = i / tr, j = i % tr;
        o = j * tK + k;
    }
    out[o] = accumulate ? out[o] + s : s;
}

// many small reductions in one launch: blockIdx.y selects the descriptor.  4 lanes share an output, 32 when the
// tensor has more than 64 partial rows (RN50 layer1: 784), combined by shuffles in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_multi_kernel(const ffm_reduce_desc* __restrict__ descs) {
    const ffm_reduce_desc d = descs[blockIdx.y];
    const int L = d.nsplit > 64 ? 32 : 4, sh = d.nsplit > 64 ? 5 : 2;
    if ((long long)blockIdx.x * blockDim.x >= (long long)L * d.n) return;     // whole block beyond this tensor (uniform)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> sh, q = gid & (L - 1);
    float s = 0.f;
    if (i < d.n)
        for (int sp = q; sp < d.nsplit; sp += L) s += d.part[(size_t)sp * d.n + i];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (L == 32) {
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
    }
    if (i >= d.n || q != 0) return;
    int o = i;
    if (d.transpose_K > 0) {
        const int k = i / d.transpose_r, j = i % d.transpose_r;
        o = j * d.transpose_K + k;
    }
    d.out[o] = s;
}

}  // namespace

static int down_valu_blocks(int M, int K, int r, int dtype) {
    const int rows = (LD_WAVES / down_kq(K, dtype == FFM_BF16 ? 2 : 4)) * (64 / down_rp(r, dtype));
    return (M + rows - 1) / rows;
}
extern "C" int ffm_lora_down_blocks(int M, int K, int r, int dtype) {
    // EXACT: the dS partial rows the BACKWARD call of ffm_lora_down (P given as [r][K] rows, t_fwd / ds_part set: the only
    // calls that write them) writes for this very (M, K, r, dtype) - the count a reduction over ds_part must use.
    if (down_mfma_ok(M, K, r, dtype, 1)) return (M + LDM_ROWS - 1) / LDM_ROWS;
    return down_valu_blocks(M, K, r, dtype);
}
extern "C" int ffm_lora_down_blocks_max(int M_max, int K, int r, int dtype) {
    // UPPER bound over every M <= M_max and both kernels, for SIZING ds_part once from the largest batch: a smaller last
    // batch (rows < 1024) falls back to the VALU kernel, which can write more partial rows per M than the matrix-core one.
    const int valu = down_valu_blocks(M_max, K, r, dtype);
    const int mf = down_mfma_ok(M_max, K, r, dtype, 1) ? (M_max + LDM_ROWS - 1) / LDM_ROWS : 0;
    return mf > valu ? mf : valu;
}
extern "C" int ffm_lora_grad_splits(int M) { return (M + LG_ROWS - 1) / LG_ROWS; }

extern "C" int ffm_lora_down(const void* x, int ldx, const float* P, int layout_rk, const float* S,
                             const int32_t* attr, int M, int K, int r, int G, int rows_per_sample, float scaling,
                             float lambda_group, float* t, float* ts, const float* t_fwd, float* ds_part, int dtype,
                             void* stream) {
    if (!x || !P || !S || M <= 0 || K <= 0 || r <= 0 || r > FFM_MAX_RANK || G <= 0 || G > FFM_MAX_GROUPS ||
        rows_per_sample <= 0)
        return FFM_EINVAL;
    if ((t_fwd == nullptr) != (ds_part == nullptr)) return FFM_EINVAL;
    if (G * r > 256) return FFM_EUNSUP;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    if (dtype != FFM_BF16 && dtype != FFM_F32) return FFM_EINVAL;
    if (((size_t)K * es) % 16 || ((size_t)ldx * es) % 16 || ((uintptr_t)x & 15)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (down_mfma_ok(M, K, r, dtype, layout_rk)) {
        hipLaunchKernelGGL(lora_down_mfma_kernel, dim3((M + LDM_ROWS - 1) / LDM_ROWS), dim3(LDM_WAVES * 64), 0, s, (const bf16_t*)x,
                           ldx, P, S, attr, M, K, r, G, rows_per_sample, scaling, lambda_group, t, ts, t_fwd, ds_part);
        FFM_CHECK_LAUNCH();
        return FFM_OK;
    }
#define DOWN(T, RP, RR, J0) launch_down<T, RP>(x, ldx, P, layout_rk, S, attr, M, K, RR, G, rows_per_sample, scaling, \
                                               lambda_group, t, ts, t_fwd, ds_part, r, J0, s)
    if (r > 16) {                       // rank 17..32: two passes over the rank (second pass re-reads x)
        int e = dtype == FFM_BF16 ? DOWN(bf16_t, 16, 16, 0) : DOWN(float, 16, 16, 0);
        if (e) return e;
        return dtype == FFM_BF16 ? DOWN(bf16_t, 16, r - 16, 16) : DOWN(float, 16, r - 16, 16);
    }
    if (dtype == FFM_BF16) return r <= 8 ? DOWN(bf16_t, 8, r, 0) : DOWN(bf16_t, 16, r, 0);
    return r <= 4 ? DOWN(float, 4, r, 0) : r <= 8 ? DOWN(float, 8, r, 0) : DOWN(float, 16, r, 0);
#undef DOWN
}

extern "C" int ffm_lora_grad_partial(const void* x, int ldx, const float* v, int M, int K, int r, float* part,
                                     int dtype, void* stream) {
    if (!x || !v || !part || M <= 0 || K <= 0 || r <= 0 || r > FFM_MAX_RANK) return FFM_EINVAL;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    if (dtype != FFM_BF16 && dtype != FFM_F32) return FFM_EINVAL;
    if (((size_t)K * es) % 16 || ((size_t)ldx * es) % 16 || ((uintptr_t)x & 15)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16 && K % LGM_COLS == 0) {             // matrix-core path (rank slots 16 at a time)
        for (int j0 = 0; j0 < r; j0 += 16) {
            const int e = launch_grad_mfma(x, ldx, v, M, K, (r - j0) < 16 ? (r - j0) : 16, part, r, j0, s);
            if (e) return e;
        }
        return FFM_OK;
    }
#define GRAD(T, RP, RR, J0) launch_grad<T, RP>(x, ldx, v, M, K, RR, part, r, J0, s)
    if (r > 16) {
        int e = dtype == FFM_BF16 ? GRAD(bf16_t, 16, 16, 0) : GRAD(float, 16, 16, 0);
        if (e) return e;
        return dtype == FFM_BF16 ? GRAD(bf16_t, 16, r - 16, 16) : GRAD(float, 16, r - 16, 16);
    }
    if (dtype == FFM_BF16) return r <= 4 ? GRAD(bf16_t, 4, r, 0) : r <= 8 ? GRAD(bf16_t, 8, r, 0) : GRAD(bf16_t, 16, r, 0);
    return r <= 4 ? GRAD(float, 4, r, 0) : r <= 8 ? GRAD(float, 8, r, 0) : GRAD(float, 16, r, 0);
#undef GRAD
}

extern "C" int ffm_lora_grad_partial_ln(const void* x, int ldx, const float* v, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, int M, int K, int r, float* part,
                                        int dtype, void* stream) {
    if (!x || !v || !mean || !rstd || !gamma || !beta || !part || M <= 0 || K <= 0 || r <= 0) return FFM_EINVAL;
    if (dtype != FFM_BF16 || K % LGM_COLS || r > 16) return FFM_EUNSUP;
    if (((size_t)ldx * 2) % 16 || ((uintptr_t)x & 15)) return FFM_EINVAL;
    const lgm_ln ln = {mean, rstd, gamma, beta};
    return launch_grad_mfma(x, ldx, v, M, K, r, part, r, 0, (hipStream_t)stream, &ln);
}

extern "C" int ffm_reduce_partials_multi(const ffm_reduce_desc* descs_dev, int ndesc, int max_n, void* stream) {
    if (!descs_dev || ndesc <= 0 || max_n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((4 * max_n + 255) / 256, ndesc), dim3(256), 0, (hipStream_t)stream,
                       descs_dev);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_reduce_partials(const float* part, int nsplit, int n, float* out, int transpose_K,
                                   int transpose_r, int accumulate, void* stream) {
    if (!part || !out || nsplit <= 0 || n <= 0) return FFM_EINVAL;
    if (transpose_K > 0 && (transpose_r <= 0 || transpose_K * transpose_r != n)) return FFM_EINVAL;
    if (nsplit >= 512 && (long long)n * 16 < (long long)nsplit * 64)          // many rows of a short tensor
        hipLaunchKernelGGL(reduce_partials_tall_kernel, dim3((n + 63) / 64), dim3(1024), 0, (hipStream_t)stream, part, nsplit, n,
                           out, transpose_K, transpose_r, accumulate);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, part,
                           nsplit, n, out, transpose_K, transpose_r, accumulate);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
