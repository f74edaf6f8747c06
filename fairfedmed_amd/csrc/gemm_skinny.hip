// Skinny GEMM: C = epilogue(A * B^T) for M <= 64 rows (the text tower: 4 prompts x 10 tokens = 40 rows against
// 512..2048-wide frozen weights).  The 128x128 kernel gives such a product N/128 = 4..16 blocks, each streaming its
// 128-column slice of the weight serially: 10-22 us of pure latency per launch, 96 launches per step on the side
// stream.  Here a block owns 16 output columns and ALL rows; its four waves split K, every wave issues the loads of 8
// K32 steps before the first MFMA, and the partial accumulators meet in LDS: N/16 = 32..128 blocks, 2-5 us.
// bf16 and f32 (the text tower always runs in f32, engine.py: class discrimination rides on the small DIFFERENCE of two
// similar prompt features, which bf16 activations blur); epilogues: none, bias, bias+residual, bias+GELU, dGELU.
#include "gemm_panel.h"

namespace {

constexpr int SK_COLS = 16, SK_WAVES = 4, SK_MF = 4, SK_UN = 8;

template <typename T, int FL>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(ffm_gemm_args p) {
    typedef typename Mma16<T>::frag_t frag_t;
    constexpr int KS = Mma16<T>::kK, KC = Elem<T>::kPerChunk;     // k per fragment step, elements per 16-byte chunk
    __shared__ f32x4 red[SK_WAVES - 1][SK_MF][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * SK_COLS;
    const int nmf = (p.M + 15) >> 4;                               // 1..4 row fragments (uniform)
    const int kw = p.K / SK_WAVES, k0 = wave * kw;                 // this wave's K range, a multiple of KS
    const T* A = reinterpret_cast<const T*>(p.a);
    const T* bp = reinterpret_cast<const T*>(p.b) + (size_t)(n0 + col) * p.ldb + k0 + kg * KC;
    const T* ap[SK_MF];
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) {
        int row = mf * 16 + col;
        row = row < p.M ? row : p.M - 1;                           // clamped rows are never stored
        ap[mf] = A + (size_t)row * p.lda + k0 + kg * KC;
    }
    f32x4 acc[SK_MF];
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < kw; ks += KS * SK_UN) {
        frag_t bf[SK_UN], af[SK_UN][SK_MF];
#pragma unroll
        for (int u = 0; u < SK_UN; ++u) {
            if (ks + KS * u < kw) {
                bf[u] = *reinterpret_cast<const frag_t*>(bp + ks + KS * u);
#pragma unroll
                for (int mf = 0; mf < SK_MF; ++mf)
                    if (mf < nmf) af[u][mf] = *reinterpret_cast<const frag_t*>(ap[mf] + ks + KS * u);
            }
        }
#pragma unroll
        for (int u = 0; u < SK_UN; ++u) {
            if (ks + KS * u < kw) {
#pragma unroll
                for (int mf = 0; mf < SK_MF; ++mf)
                    if (mf < nmf) Mma16<T>::mma(acc[mf], af[u][mf], bf[u]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mf = 0; mf < SK_MF; ++mf)
            if (mf < nmf) red[wave - 1][mf][lane] = acc[mf];
    }
    __syncthreads();
    if (wave != 0) return;
    const int n = n0 + col;
    const float bias = (FL & FFM_EPI_BIAS) ? p.bias[n] : 0.f;
    T* C = reinterpret_cast<T*>(p.c);
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) {
        if (mf >= nmf) break;
        f32x4 v = acc[mf];
#pragma unroll
        for (int w = 0; w < SK_WAVES - 1; ++w) {
            const f32x4 o = red[w][mf][lane];
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = mf * 16 + 4 * kg + e;
            if (row >= p.M) continue;
            const size_t o = (size_t)row * p.ldc + n;
            float x = v[e] + bias;
            if (FL & FFM_EPI_RESIDUAL) x += (float)reinterpret_cast<const T*>(p.res)[o];
            if (FL & FFM_EPI_DGELU) x *= Act<T>::gelu_grad((float)reinterpret_cast<const T*>(p.aux)[o]);
            C[o] = (T)x;
            if (FL & FFM_EPI_GELU) reinterpret_cast<T*>(p.c2)[o] = (T)Act<T>::gelu(x);
        }
    }
}

template <typename T, int FL>
int launch(const ffm_gemm_args& a, hipStream_t s) {
    hipLaunchKernelGGL((gemm_skinny_kernel<T, FL>), dim3(a.N / SK_COLS), dim3(SK_WAVES * 64), 0, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

template <typename T>
int launch_flags(const ffm_gemm_args& a, hipStream_t s) {
    switch (a.flags) {
        case 0: return launch<T, 0>(a, s);
        case FFM_EPI_BIAS: return launch<T, FFM_EPI_BIAS>(a, s);
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL: return launch<T, FFM_EPI_BIAS | FFM_EPI_RESIDUAL>(a, s);
        case FFM_EPI_BIAS | FFM_EPI_GELU: return launch<T, FFM_EPI_BIAS | FFM_EPI_GELU>(a, s);
        case FFM_EPI_DGELU: return launch<T, FFM_EPI_DGELU>(a, s);
    }
    return FFM_EINVAL;
}

}  // namespace

bool ffm_skinny_ok(const ffm_gemm_args& a, int dtype) {
    if ((dtype != FFM_BF16 && dtype != FFM_F32) || a.M > 16 * SK_MF || a.N % SK_COLS || a.K % (32 * SK_WAVES)) return false;
    switch (a.flags) {
        case 0:
        case FFM_EPI_BIAS:
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL:
        case FFM_EPI_BIAS | FFM_EPI_GELU:
        case FFM_EPI_DGELU: return true;
    }
    return false;
}

int ffm_skinny_launch(const ffm_gemm_args& a, int dtype, hipStream_t s) {
    return dtype == FFM_F32 ? launch_flags<float>(a, s) : launch_flags<bf16_t>(a, s);
}
