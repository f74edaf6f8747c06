// Rank-r FairLoRA kernels (HBM-bound, VALU + LDS + wave shuffles; no MFMA):
//   lora_down         t = x P (P staged in LDS tiles), ts = scaling * t * s_b, dS partials
//   lora_grad_partial part[s] = sum_{rows of split s} x^T v   (dA / dB without forming dW)
//   reduce_partials   deterministic second stage
#include "common.h"

namespace {

// ---------------------------------------------------------------------------
// lora_down: block = 4 waves x 8 rows; lane -> (row = lane & 7, kgroup = lane >> 3).
// Per step a wave reads 8 rows x 128 contiguous bytes (full lines); the K-tile
// of P lives in LDS as Ps[k][RP] and lanes of one k-group broadcast-read it.
// ---------------------------------------------------------------------------
constexpr int LD_ROWS = 32;     // rows per block
constexpr int LD_NIT = 8;       // 16-byte loads in flight per lane per K-tile

template <typename T, int RP>
__global__ __launch_bounds__(256) void lora_down_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ P,
                                                        int layout_rk, const float* __restrict__ S,
                                                        const int32_t* __restrict__ attr, int M, int K, int r, int G,
                                                        int rows_per_sample, float scaling, float lambda_group,
                                                        float* __restrict__ t_out, float* __restrict__ ts_out,
                                                        const float* __restrict__ t_fwd, float* __restrict__ ds_part) {
    constexpr int CE = Elem<T>::kPerChunk;          // elements per 16 B
    constexpr int KSTEP = 8 * CE;                    // k covered by one wave step
    constexpr int KT = LD_NIT * KSTEP;               // k per LDS tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ps = reinterpret_cast<float*>(smem);      // [KT][RP]
    float* Vs = Ps + KT * RP;                        // [32][RP] dS staging

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsub = lane & 7, kg = lane >> 3;
    const int row = blockIdx.x * LD_ROWS + wave * 8 + rsub;
    const int lrow = row < M ? row : M - 1;
    const T* xr = x + (size_t)lrow * ldx;

    float acc[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) acc[j] = 0.f;

    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kt = (K - k0) < KT ? (K - k0) : KT;
        __syncthreads();
        // stage P[k0 .. k0+kt) -> Ps[kk][j], zero-padding j >= r
        for (int idx = tid; idx < kt * RP; idx += 256) {
            float v = 0.f;
            if (layout_rk) {
                const int j = idx / kt, kk = idx % kt;       // read along k (contiguous in P[j][:])
                if (j < r) v = P[(size_t)j * K + k0 + kk];
                Ps[kk * RP + j] = v;
            } else {
                const int kk = idx / RP, j = idx % RP;
                if (j < r) v = P[(size_t)(k0 + kk) * r + j];
                Ps[idx] = v;
            }
        }
        // issue this lane's loads for the tile
        typename Elem<T>::chunk_t xv[LD_NIT];
#pragma unroll
        for (int it = 0; it < LD_NIT; ++it) {
            const int k = k0 + it * KSTEP + kg * CE;
            if (k < K) xv[it] = *reinterpret_cast<const typename Elem<T>::chunk_t*>(xr + k);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LD_NIT; ++it) {
            const int kk0 = it * KSTEP + kg * CE;
            if (k0 + kk0 < K) {
#pragma unroll
                for (int e = 0; e < CE; ++e) {
                    const float xe = Elem<T>::to_f(xv[it][e]);
                    const float* pr = Ps + (kk0 + e) * RP;
#pragma unroll
                    for (int j4 = 0; j4 < RP / 4; ++j4) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pr + j4 * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[j4 * 4 + c] += xe * pv[c];
                    }
                }
            }
        }
    }
    // reduce across the 8 k-groups (lanes differing in bits 3..5)
#pragma unroll
    for (int j = 0; j < RP; ++j) {
        float v = acc[j];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[j] = v;
    }
    const bool writer = (kg == 0) && (row < M);
    const int sample = lrow / rows_per_sample;
    if (writer) {
        for (int j = 0; j < r; ++j) {
            float sb = 0.f;
            for (int g = 0; g < G; ++g) sb += group_mix_w(attr, sample, g, G, lambda_group) * S[g * r + j];
            if (t_out) t_out[(size_t)row * r + j] = acc[j];
            if (ts_out) ts_out[(size_t)row * r + j] = scaling * acc[j] * sb;
        }
    }
    if (t_fwd && ds_part) {
        __syncthreads();
        if (kg == 0) {
            for (int j = 0; j < r; ++j)
                Vs[(wave * 8 + rsub) * RP + j] = (row < M) ? scaling * t_fwd[(size_t)row * r + j] * acc[j] : 0.f;
        }
        __syncthreads();
        if (tid < G * r) {
            const int g = tid / r, j = tid % r;
            float s = 0.f;
            for (int rr = 0; rr < LD_ROWS; ++rr) {
                const int grow = blockIdx.x * LD_ROWS + rr;
                if (grow < M) s += group_mix_w(attr, grow / rows_per_sample, g, G, lambda_group) * Vs[rr * RP + j];
            }
            ds_part[((size_t)blockIdx.x * G + g) * r + j] = s;
        }
    }
}

template <typename T, int RP>
int launch_down(const void* x, int ldx, const float* P, int layout_rk, const float* S, const int32_t* attr, int M,
                int K, int r, int G, int rps, float scaling, float lam, float* t, float* ts, const float* t_fwd,
                float* ds_part, hipStream_t s) {
    constexpr int KT = LD_NIT * 8 * Elem<T>::kPerChunk;
    const int lds = (KT * RP + LD_ROWS * RP) * 4;
    const int blocks = (M + LD_ROWS - 1) / LD_ROWS;
    hipLaunchKernelGGL((lora_down_kernel<T, RP>), dim3(blocks), dim3(256), lds, s, (const T*)x, ldx, P, layout_rk, S,
                       attr, M, K, r, G, rps, scaling, lam, t, ts, t_fwd, ds_part);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// ---------------------------------------------------------------------------
// lora_grad_partial: one wave per (256-column group, 64-row split); lane owns 4
// consecutive columns, v[m][:] is wave-uniform (scalar loads).
// ---------------------------------------------------------------------------
constexpr int LG_ROWS = 64;

template <typename T, int RP>
__global__ __launch_bounds__(64) void lora_grad_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ v,
                                                       int M, int K, int r, float* __restrict__ part) {
    const int lane = threadIdx.x;
    const int k = blockIdx.x * 256 + lane * 4;
    const int split = blockIdx.y;
    const int m0 = split * LG_ROWS;
    const int m1 = (m0 + LG_ROWS) < M ? (m0 + LG_ROWS) : M;
    const bool active = k < K;
    float acc[4][RP];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < RP; ++j) acc[e][j] = 0.f;

    const T* xc = x + (active ? k : 0);
    for (int m = m0; m < m1; m += 8) {
        f32x4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int mm = (m + u) < m1 ? (m + u) : (m1 - 1);
            xv[u] = Vec4<T>::load(xc + (size_t)mm * ldx);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (m + u < m1) {
                const float* vr = v + (size_t)(m + u) * r;   // wave-uniform address
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    const float vj = (j < r) ? vr[j] : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e][j] += xv[u][e] * vj;
                }
            }
        }
    }
    if (active) {
        float* dst = part + ((size_t)split * K + k) * r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            for (int j = 0; j < r; ++j) dst[e * r + j] = acc[e][j];
    }
}

template <typename T, int RP>
int launch_grad(const void* x, int ldx, const float* v, int M, int K, int r, float* part, hipStream_t s) {
    dim3 grid((K + 255) / 256, (M + LG_ROWS - 1) / LG_ROWS);
    hipLaunchKernelGGL((lora_grad_kernel<T, RP>), grid, dim3(64), 0, s, (const T*)x, ldx, v, M, K, r, part);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nsplit, int n,
                                                              float* __restrict__ out, int tK, int tr,
                                                              int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * n + i];
    int o = i;
    if (tK > 0) {                       // i = k * r + j  ->  j * K + k
        const int k = i / tr, j = i % tr;
        o = j * tK + k;
    }
    out[o] = accumulate ? out[o] + s : s;
}

#define DISPATCH_RP(FN, T, r, ...)                                  \
    ((r) <= 4 ? FN<T, 4>(__VA_ARGS__) : (r) <= 8 ? FN<T, 8>(__VA_ARGS__) \
     : (r) <= 16 ? FN<T, 16>(__VA_ARGS__) : FN<T, 32>(__VA_ARGS__))

}  // namespace

extern "C" int ffm_lora_down_blocks(int M) { return (M + LD_ROWS - 1) / LD_ROWS; }
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
    if (((size_t)K * es) % 128 || ((size_t)ldx * es) % 16 || ((uintptr_t)x & 15)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        return DISPATCH_RP(launch_down, bf16_t, r, x, ldx, P, layout_rk, S, attr, M, K, r, G, rows_per_sample, scaling,
                           lambda_group, t, ts, t_fwd, ds_part, s);
    return DISPATCH_RP(launch_down, float, r, x, ldx, P, layout_rk, S, attr, M, K, r, G, rows_per_sample, scaling,
                       lambda_group, t, ts, t_fwd, ds_part, s);
}

extern "C" int ffm_lora_grad_partial(const void* x, int ldx, const float* v, int M, int K, int r, float* part,
                                     int dtype, void* stream) {
    if (!x || !v || !part || M <= 0 || K <= 0 || (K & 3) || r <= 0 || r > FFM_MAX_RANK) return FFM_EINVAL;
    const size_t es = dtype == FFM_BF16 ? 2 : 4;
    if (dtype != FFM_BF16 && dtype != FFM_F32) return FFM_EINVAL;
    if (((size_t)ldx * es) % 8 || ((uintptr_t)x & 15)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16) return DISPATCH_RP(launch_grad, bf16_t, r, x, ldx, v, M, K, r, part, s);
    return DISPATCH_RP(launch_grad, float, r, x, ldx, v, M, K, r, part, s);
}

extern "C" int ffm_reduce_partials(const float* part, int nsplit, int n, float* out, int transpose_K,
                                   int transpose_r, int accumulate, void* stream) {
    if (!part || !out || nsplit <= 0 || n <= 0) return FFM_EINVAL;
    if (transpose_K > 0 && (transpose_r <= 0 || transpose_K * transpose_r != n)) return FFM_EINVAL;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, nsplit,
                       n, out, transpose_K, transpose_r, accumulate);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
