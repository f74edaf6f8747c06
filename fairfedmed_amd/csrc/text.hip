// The two ends of the text tower on the device (float32; n_text = N prompts x n_cls classes rows, a few KB each):
//
//   ffm_text_embed     prompts = [prefix, ctx, suffix] + positional embedding  (PromptLearner.forward, class token
//                      position 'end': trainers/GLP_OT_SVLoRA.py:131-152; TextEncoder.forward :57)
//   ffm_text_tail_fwd  EOT gather -> ln_final -> @ text_projection (:62-64) -> F.normalize (:716-717) -> mean over the
//                      prompts (:713 with the OT = 'None' head) or every prompt's normalised feature (transport heads)
//   ffm_text_tail_bwd  the way back to the tower's output gradient (zero on every row but the EOT rows)
//   ffm_text_ctx_grad  d ctx = the input gradient's ctx rows summed over the classes (ctx is shared by the classes, :133-136)
//
// These were ~35 PyTorch launches per step (cat / gather / layer_norm / hipBLASLt GEMMs on 4 rows / autograd); they are 6
// launches now and nothing but HIP kernels of this library is left in a training step.
#include "common.h"

namespace {

constexpr int TT = 256;                     // threads per block
constexpr int TCOLS = 64;                   // output columns (forward) / input features (backward) per block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the block (TT threads); red: TT / 64 floats of LDS; every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TT / 64; ++i) s += red[i];
    return s;
}

// row (p, t) of the tower input: t = 0 the SOS embedding, 1..n_ctx the learned context of prompt p / n_cls, then the
// class-name tokens ... (token_suffix), all + positional_embedding[t]
__global__ __launch_bounds__(TT) void text_embed_kernel(const float* __restrict__ prefix, const float* __restrict__ ctx,
                                                        const float* __restrict__ suffix, int suffix_rows,
                                                        const float* __restrict__ pos, float* __restrict__ x, int n_cls,
                                                        int n_ctx, int TL, int w) {
    const int row = blockIdx.x, p = row / TL, t = row % TL;
    const float* src = t == 0 ? prefix + (size_t)p * w
                     : t <= n_ctx ? ctx + ((size_t)(p / n_cls) * n_ctx + (t - 1)) * w
                                  : suffix + ((size_t)p * suffix_rows + (t - 1 - n_ctx)) * w;
    for (int i = threadIdx.x; i < w; i += TT) x[(size_t)row * w + i] = src[i] + pos[(size_t)t * w + i];
}

// grid (n_text, D / TCOLS): tf[p][d0 .. d0 + 63] = LayerNorm(x[eot_row[p]]) . proj[:, d0 .. d0 + 63]
__global__ __launch_bounds__(TT) void text_tail_proj_kernel(const float* __restrict__ x, const int* __restrict__ eot_row,
                                                            const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                            const float* __restrict__ proj, float* __restrict__ tf,
                                                            float* __restrict__ stats, int w, int D) {
    extern __shared__ float sm[];
    float* y = sm;                           // [w]
    float* red = sm + w;                     // [TT / 64]
    float* part = red + TT / 64;             // [4][TCOLS]
    const int p = blockIdx.x, d0 = blockIdx.y * TCOLS, tid = threadIdx.x;
    const float* xr = x + (size_t)eot_row[p] * w;
    float s = 0.f;
    for (int i = tid; i < w; i += TT) {
        y[i] = xr[i];
        s += y[i];
    }
    const float mean = block_sum(s, red) / (float)w;
    float q = 0.f;
    for (int i = tid; i < w; i += TT) {
        const float d = y[i] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)w + 1e-5f);
    for (int i = tid; i < w; i += TT) y[i] = (y[i] - mean) * rstd * lnw[i] + lnb[i];
    if (blockIdx.y == 0 && tid == 0) {
        stats[2 * p] = mean;
        stats[2 * p + 1] = rstd;
    }
    __syncthreads();
    const int col = tid & (TCOLS - 1), ks = tid / TCOLS;      // 4 slices of the contraction, summed in a fixed order
    float acc = 0.f;
    if (d0 + col < D) {
        // 16 loads of the projection in flight per thread (a load -> FMA chain of w / 4 dependent steps took 53 us)
        constexpr int KU = 16, KSTEP = TT / TCOLS;
        int k = ks;
        for (; k + (KU - 1) * KSTEP < w; k += KU * KSTEP) {
            float pv[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) pv[u] = proj[(size_t)(k + u * KSTEP) * D + d0 + col];
#pragma unroll
            for (int u = 0; u < KU; ++u) acc += y[k + u * KSTEP] * pv[u];
        }
        for (; k < w; k += KSTEP) acc += y[k] * proj[(size_t)k * D + d0 + col];
    }
    part[ks * TCOLS + col] = acc;
    __syncthreads();
    if (tid < TCOLS && d0 + tid < D)
        tf[(size_t)p * D + d0 + tid] = (part[tid] + part[TCOLS + tid]) + (part[2 * TCOLS + tid] + part[3 * TCOLS + tid]);
}

// grid n_cls: F.normalize of every prompt's feature (tn, 1 / max(norm, 1e-12) kept for the way back) and, with `tbar`,
// their mean over the prompts
__global__ __launch_bounds__(TT) void text_tail_norm_kernel(const float* __restrict__ tf, float* __restrict__ tn,
                                                            float* __restrict__ rnorm, float* __restrict__ tbar,
                                                            int n_prompts, int n_cls, int D) {
    __shared__ float red[TT / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    for (int n = 0; n < n_prompts; ++n) {
        const int p = n * n_cls + c;
        float q = 0.f;
        for (int i = tid; i < D; i += TT) {
            const float v = tf[(size_t)p * D + i];
            q += v * v;
        }
        const float rn = 1.0f / fmaxf(sqrtf(block_sum(q, red)), 1e-12f);
        if (tid == 0) rnorm[p] = rn;
        for (int i = tid; i < D; i += TT) tn[(size_t)p * D + i] = tf[(size_t)p * D + i] * rn;
    }
    if (tbar) {
        __syncthreads();
        for (int i = tid; i < D; i += TT) {
            float s = 0.f;
            for (int n = 0; n < n_prompts; ++n) s += tn[(size_t)(n * n_cls + c) * D + i];     // (this thread's own writes)
            tbar[(size_t)c * D + i] = s / (float)n_prompts;
        }
    }
}

// grid (n_text, w / TCOLS): d tn -> d tf (normalize backward) -> dy[p][k0 .. k0 + 63] = d tf . proj[k]^T
__global__ __launch_bounds__(TT) void text_tail_bwd_proj_kernel(const float* __restrict__ tn, const float* __restrict__ rnorm,
                                                                const float* __restrict__ dtbar, const float* __restrict__ dtn,
                                                                const float* __restrict__ proj, float* __restrict__ dy,
                                                                int n_prompts, int n_cls, int w, int D) {
    extern __shared__ float sm[];
    float* dtf = sm;                         // [D]
    float* red = sm + D;
    const int p = blockIdx.x, k0 = blockIdx.y * TCOLS, tid = threadIdx.x, c = p % n_cls;
    const float inv_n = 1.0f / (float)n_prompts;
    float dot = 0.f;
    for (int i = tid; i < D; i += TT) {
        const float g = dtn ? dtn[(size_t)p * D + i] : dtbar[(size_t)c * D + i] * inv_n;
        dtf[i] = g;
        dot += g * tn[(size_t)p * D + i];
    }
    dot = block_sum(dot, red);
    const float rn = rnorm[p];
    for (int i = tid; i < D; i += TT) dtf[i] = (dtf[i] - tn[(size_t)p * D + i] * dot) * rn;
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    // a wave per input feature (a coalesced row of proj), four features at a time so that their loads are in flight
    // together (one feature after the other: 57 us)
    constexpr int KB = 4;
    for (int kk = wv * KB; kk < TCOLS; kk += (TT / 64) * KB) {
        float acc[KB];
        const float* row[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            acc[u] = 0.f;
            const int k = k0 + kk + u < w ? k0 + kk + u : w - 1;         // (clamped: computed, not stored)
            row[u] = proj + (size_t)k * D;
        }
        constexpr int IU = 4;                                            // KB x IU = 16 loads in flight per lane
        int i = lane;
        for (; i + (IU - 1) * 64 < D; i += IU * 64) {
            float pv[KB][IU];
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int v = 0; v < IU; ++v) pv[u][v] = row[u][i + v * 64];
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int v = 0; v < IU; ++v) acc[u] += dtf[i + v * 64] * pv[u][v];
        }
        for (; i < D; i += 64)
#pragma unroll
            for (int u = 0; u < KB; ++u) acc[u] += dtf[i] * row[u][i];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const float t = wave_sum(acc[u]);
            if (lane == 0 && k0 + kk + u < w) dy[(size_t)p * w + k0 + kk + u] = t;
        }
    }
}

// grid rows (n_text * TL): the EOT row of prompt p receives the LayerNorm backward of dy[p]; every other row is zero
__global__ __launch_bounds__(TT) void text_tail_bwd_ln_kernel(const float* __restrict__ x, const int* __restrict__ eot_row,
                                                              const float* __restrict__ lnw, const float* __restrict__ stats,
                                                              const float* __restrict__ dy, float* __restrict__ g, int TL, int w) {
    __shared__ float red[TT / 64];
    const int row = blockIdx.x, p = row / TL, tid = threadIdx.x;
    float* gr = g + (size_t)row * w;
    if (eot_row[p] != row) {
        for (int i = tid; i < w; i += TT) gr[i] = 0.f;
        return;
    }
    const float mean = stats[2 * p], rstd = stats[2 * p + 1];
    const float* xr = x + (size_t)row * w;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < w; i += TT) {
        const float a = dy[(size_t)p * w + i] * lnw[i], xh = (xr[i] - mean) * rstd;
        s1 += a;
        s2 += a * xh;
    }
    const float m1 = block_sum(s1, red) / (float)w;
    const float m2 = block_sum(s2, red) / (float)w;
    for (int i = tid; i < w; i += TT) {
        const float a = dy[(size_t)p * w + i] * lnw[i], xh = (xr[i] - mean) * rstd;
        gr[i] = rstd * (a - m1 - xh * m2);
    }
}

// grid (N * n_ctx): d ctx[n][j] = sum_c g[(n n_cls + c) TL + 1 + j]
__global__ __launch_bounds__(TT) void text_ctx_grad_kernel(const float* __restrict__ g, float* __restrict__ dctx, int n_cls,
                                                           int n_ctx, int TL, int w) {
    const int n = blockIdx.x / n_ctx, j = blockIdx.x % n_ctx;
    for (int i = threadIdx.x; i < w; i += TT) {
        float s = 0.f;
        for (int c = 0; c < n_cls; ++c) s += g[((size_t)(n * n_cls + c) * TL + 1 + j) * w + i];
        dctx[((size_t)n * n_ctx + j) * w + i] = s;
    }
}

}  // namespace

extern "C" int ffm_text_embed(const float* prefix, const float* ctx, const float* suffix, int suffix_rows, const float* pos,
                              float* x, int n_prompts, int n_cls, int n_ctx, int TL, int w, void* stream) {
    if (!prefix || !ctx || !suffix || !pos || !x || n_prompts <= 0 || n_cls <= 0 || n_ctx <= 0 || w <= 0) return FFM_EINVAL;
    if (TL < 1 + n_ctx || TL - 1 - n_ctx > suffix_rows) return FFM_EINVAL;
    hipLaunchKernelGGL(text_embed_kernel, dim3(n_prompts * n_cls * TL), dim3(TT), 0, (hipStream_t)stream, prefix, ctx, suffix,
                       suffix_rows, pos, x, n_cls, n_ctx, TL, w);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_text_tail_fwd(const float* x, const int* eot_row, const float* lnw, const float* lnb, const float* proj,
                                 float* tf, float* tn, float* rnorm, float* stats, float* tbar, int n_prompts, int n_cls, int w,
                                 int D, void* stream) {
    if (!x || !eot_row || !lnw || !lnb || !proj || !tf || !tn || !rnorm || !stats) return FFM_EINVAL;
    if (n_prompts <= 0 || n_cls <= 0 || w <= 0 || D <= 0 || w > 8192) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int n_text = n_prompts * n_cls;
    hipLaunchKernelGGL(text_tail_proj_kernel, dim3(n_text, (D + TCOLS - 1) / TCOLS), dim3(TT),
                       (w + TT / 64 + 4 * TCOLS) * sizeof(float), s, x, eot_row, lnw, lnb, proj, tf, stats, w, D);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(text_tail_norm_kernel, dim3(n_cls), dim3(TT), 0, s, tf, tn, rnorm, tbar, n_prompts, n_cls, D);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_text_tail_bwd(const float* x, const int* eot_row, const float* lnw, const float* proj, const float* tn,
                                 const float* rnorm, const float* stats, const float* dtbar, const float* dtn, float* dy,
                                 float* g, int n_prompts, int n_cls, int TL, int w, int D, void* stream) {
    if (!x || !eot_row || !lnw || !proj || !tn || !rnorm || !stats || !dy || !g) return FFM_EINVAL;
    if ((dtbar == nullptr) == (dtn == nullptr)) return FFM_EINVAL;        // exactly one of the two incoming gradients
    if (n_prompts <= 0 || n_cls <= 0 || TL <= 0 || w <= 0 || D <= 0 || D > 8192) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int n_text = n_prompts * n_cls;
    hipLaunchKernelGGL(text_tail_bwd_proj_kernel, dim3(n_text, (w + TCOLS - 1) / TCOLS), dim3(TT), (D + TT / 64) * sizeof(float),
                       s, tn, rnorm, dtbar, dtn, proj, dy, n_prompts, n_cls, w, D);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(text_tail_bwd_ln_kernel, dim3(n_text * TL), dim3(TT), 0, s, x, eot_row, lnw, stats, dy, g, TL, w);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_text_ctx_grad(const float* g, float* dctx, int n_prompts, int n_cls, int n_ctx, int TL, int w, void* stream) {
    if (!g || !dctx || n_prompts <= 0 || n_cls <= 0 || n_ctx <= 0 || TL < 1 + n_ctx || w <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(text_ctx_grad_kernel, dim3(n_prompts * n_ctx), dim3(TT), 0, (hipStream_t)stream, g, dctx, n_cls, n_ctx, TL, w);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
