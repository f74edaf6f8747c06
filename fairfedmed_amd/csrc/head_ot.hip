// Logits heads with an optimal-transport plan (TRAINER.GLP_OT.OT = 'Sinkhorn' | 'COT'; SURVEY.md §8 a15 / (f)-4;
// trainers/GLP_OT_SVLoRA.py:615-675, 713-757).  For every (image b, class c):
//   sim[m][n] = <f^[b,m], t^[n,c]>              m = 1..L-1 image tokens (the class token is dropped), n = prompts
//   K = exp(-(1 - sim) / eps); the plan T = scaling iterations on K (under no_grad in the reference: T is a constant
//   of the backward pass); logits[b][c] = exp(logit_scale) * sum_{m,n} T[m][n] sim[m][n].
// The reference stops ALL problems of the batch at the same iteration (one scalar test on the batch mean of the
// iterate's change, with a host sync per iteration).  Here: pass 1 runs max_iter iterations per problem and records
// each iteration's local change, a one-block kernel finds the first iteration whose batch mean is below the
// threshold, pass 2 re-runs exactly that many iterations and emits T and the logits - no host sync, no grid barrier.
#include "common.h"

namespace {

constexpr int OT_MAXM = 256, OT_MAXN = 8, OT_MAXC = 8, OT_MAXV = 4;    // tokens, prompts, classes, D <= 1024
constexpr int OT_MAXP = OT_MAXN * OT_MAXC;

// one wave per token row: rnorm and the P = n_cls * N cosine similarities of the row
template <typename T>
__global__ __launch_bounds__(256) void ot_sim_kernel(const T* __restrict__ f, const float* __restrict__ tn,
                                                     float* __restrict__ rnorm, float* __restrict__ sim, int B, int L, int D,
                                                     int n_cls, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= B * L) return;
    const int b = row / L, l = row % L, M = L - 1, nchunk = D >> 2;
    if (l == 0) {
        if (lane == 0) rnorm[row] = 0.f;
        return;
    }
    f32x4 v[OT_MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < OT_MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) {
            v[i] = Vec4<T>::load(f + (size_t)row * D + c * 4);
            ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
    }
    const float rn = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);              // F.normalize eps
    if (lane == 0) rnorm[row] = rn;
    for (int c = 0; c < n_cls; ++c)
        for (int n = 0; n < N; ++n) {
            const float* t = tn + (size_t)(n * n_cls + c) * D;
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < OT_MAXV; ++i) {
                const int ch = lane + 64 * i;
                if (ch < nchunk) {
                    const f32x4 tv = *reinterpret_cast<const f32x4*>(t + ch * 4);
                    d += v[i][0] * tv[0] + v[i][1] * tv[1] + v[i][2] * tv[2] + v[i][3] * tv[3];
                }
            }
            d = wave_sum(d) * rn;
            if (lane == 0) sim[(((size_t)b * n_cls + c) * M + (l - 1)) * N + n] = d;
        }
}

// block = one (b, c) problem, thread = one token m.  MODE 1: Sinkhorn (iterate r, c; change of r), MODE 2: COT
// (iterate u, v; change of v).  PASS 1 records errs[it][prob]; PASS 2 stops after *istop + 1 iterations and writes T
// and the problem's sum of T * sim.
template <int MODE, int PASS>
__global__ __launch_bounds__(OT_MAXM) void ot_plan_kernel(const float* __restrict__ sim, float* __restrict__ errs,
                                                          const int32_t* __restrict__ istop, float* __restrict__ Tout,
                                                          float* __restrict__ tsum, int M, int N, int nprob, float eps,
                                                          float top, int max_iter) {
    __shared__ float red[OT_MAXM / 64][OT_MAXN + 1];
    __shared__ float col[OT_MAXN];                                   // c (Sinkhorn) or v (COT)
    const int p = blockIdx.x, m = threadIdx.x, lane = m & 63, wave = m >> 6, nw = (blockDim.x + 63) >> 6;
    const bool live = m < M;
    float K[OT_MAXN], s[OT_MAXN];
#pragma unroll
    for (int n = 0; n < OT_MAXN; ++n) {
        s[n] = (live && n < N) ? sim[((size_t)p * M + m) * N + n] : 0.f;
        K[n] = (live && n < N) ? expf(-(1.0f - s[n]) / eps) : 0.f;
    }
    const float a = 1.0f / (float)M;                                 // xx = 1 / M
    const float bm = (MODE == 1 ? 1.0f : top) / (float)N;       // yy (COT: * min(sum(xx) = #problems, TOP_PERCENT), host)
    if (m < OT_MAXN) col[m] = 1.0f;
    float rv = 1.0f;                                                 // r (Sinkhorn) or u (COT) of this token
    __syncthreads();
    const int iters = PASS == 1 ? max_iter : (*istop + 1);
    for (int it = 0; it < iters; ++it) {
        float dot = 0.f;
#pragma unroll
        for (int n = 0; n < OT_MAXN; ++n)
            if (n < N) dot += K[n] * col[n];
        float change = 0.f, rnew;
        if (MODE == 1) {
            rnew = a / dot;                                          // r = u / (K c)
            change = live ? fabsf(rnew - rv) : 0.f;
        } else {
            rnew = fminf(1.0f / (dot / a), 1.0f);                    // u = min(dx / (Kp v), dx), Kp = K / a
        }
        rv = live ? rnew : 0.f;
        // column sums sum_m K[m][n] * r[m]  (+ the change of r for Sinkhorn): wave shuffles, then across the waves
        float part[OT_MAXN + 1];
#pragma unroll
        for (int n = 0; n < OT_MAXN; ++n) part[n] = n < N ? wave_sum(K[n] * rv) : 0.f;
        part[OT_MAXN] = wave_sum(change);
        __syncthreads();                                             // everyone has read col[] of this iteration
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n <= OT_MAXN; ++n) red[wave][n] = part[n];
        }
        __syncthreads();
        if (m <= OT_MAXN) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[w][m];
            if (m < OT_MAXN) {
                if (m < N) {
                    const float cn = MODE == 1 ? bm / t : 1.0f / (t / bm);   // c = v / (K^T r);  v = dy / (Kq u), Kq = K^T / b
                    if (MODE == 2 && PASS == 1) red[0][m] = fabsf(cn - col[m]);   // change of v (own slot, read below)
                    col[m] = cn;
                }
            } else if (MODE == 1 && PASS == 1) {
                errs[(size_t)it * nprob + p] = t;                    // sum_m |r - r0|
            }
        }
        __syncthreads();
        if (MODE == 2 && PASS == 1 && m == 0) {
            float t = 0.f;
            for (int n = 0; n < N; ++n) t += red[0][n];
            errs[(size_t)it * nprob + p] = t;                        // sum_n |v - v0|
        }
        if (MODE == 2) __syncthreads();
    }
    if (PASS == 2) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < OT_MAXN; ++n) {
            if (live && n < N) {
                const float t = rv * K[n] * col[n];                  // T = diag(r) K diag(c)
                Tout[((size_t)p * M + m) * N + n] = t;
                acc += t * s[n];
            }
        }
        acc = wave_sum(acc);
        __syncthreads();
        if (lane == 0) red[wave][0] = acc;
        __syncthreads();
        if (m == 0) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[w][0];
            tsum[p] = t;
        }
    }
}

// first iteration whose batch-mean change is below the threshold (else the last one); logits of pass 2 are scaled later
__global__ __launch_bounds__(256) void ot_stop_kernel(const float* __restrict__ errs, int32_t* __restrict__ istop, int nprob,
                                                      int max_iter, float denom, float thresh) {
    __shared__ float red[4];
    __shared__ int found;
    if (threadIdx.x == 0) found = max_iter - 1;
    __syncthreads();
    for (int it = 0; it < max_iter; ++it) {
        float t = 0.f;
        for (int p = threadIdx.x; p < nprob; p += 256) t += errs[(size_t)it * nprob + p];
        t = wave_sum(t);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / denom;
        __syncthreads();
        if (mean < thresh) {
            if (threadIdx.x == 0) found = it;
            break;                                                   // uniform: every thread sees the same mean
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *istop = found;
}

__global__ void ot_logits_kernel(const float* __restrict__ tsum, const float* __restrict__ logit_scale,
                                 float* __restrict__ logits, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) logits[i] = expf(logit_scale[0]) * tsum[i];
}

// backward.  One wave per token row: d f^ = sum_{c,n} w[c][n] t^[n,c], w = dlogit[b][c] e^{ls} T[b,c,m,n];
// d f = rn (d f^ - f^ <f^, d f^>).  Class-token rows get zeros.
template <typename T>
__global__ __launch_bounds__(256) void ot_bwd_feat_kernel(const T* __restrict__ f, const float* __restrict__ tn,
                                                          const float* __restrict__ logit_scale,
                                                          const float* __restrict__ rnorm, const float* __restrict__ Tp,
                                                          const float* __restrict__ dlogits, T* __restrict__ df, int B, int L,
                                                          int D, int n_cls, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= B * L) return;
    const int b = row / L, l = row % L, M = L - 1, nchunk = D >> 2;
    f32x4 g[OT_MAXV];
#pragma unroll
    for (int i = 0; i < OT_MAXV; ++i) g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (l > 0) {
        const float es = expf(logit_scale[0]), rn = rnorm[row];
        f32x4 fh[OT_MAXV];
#pragma unroll
        for (int i = 0; i < OT_MAXV; ++i) {
            const int c = lane + 64 * i;
            fh[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < nchunk) {
                fh[i] = Vec4<T>::load(f + (size_t)row * D + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) fh[i][e] *= rn;
            }
        }
        for (int c = 0; c < n_cls; ++c)
            for (int n = 0; n < N; ++n) {
                const float w = dlogits[b * n_cls + c] * es * Tp[(((size_t)b * n_cls + c) * M + (l - 1)) * N + n];
                const float* t = tn + (size_t)(n * n_cls + c) * D;
#pragma unroll
                for (int i = 0; i < OT_MAXV; ++i) {
                    const int ch = lane + 64 * i;
                    if (ch < nchunk) {
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(t + ch * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[i][e] += w * tv[e];
                    }
                }
            }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < OT_MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot += fh[i][e] * g[i][e];
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < OT_MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[i][e] = rn * (g[i][e] - fh[i][e] * dot);
    }
#pragma unroll
    for (int i = 0; i < OT_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) Vec4<T>::store(df + (size_t)row * D + c * 4, g[i]);
    }
}

// d t^[n,c] partials per image: part[b][n * n_cls + c][d] = sum_m w[b,c,m,n] f^[b,m,d]   (reduced over b afterwards)
template <typename T>
__global__ __launch_bounds__(256) void ot_bwd_text_kernel(const T* __restrict__ f, const float* __restrict__ logit_scale,
                                                          const float* __restrict__ rnorm, const float* __restrict__ Tp,
                                                          const float* __restrict__ dlogits, float* __restrict__ part, int L,
                                                          int D, int n_cls, int N) {
    const int b = blockIdx.x, pc = blockIdx.y, n = pc / n_cls, c = pc % n_cls, M = L - 1;
    const float gs = dlogits[b * n_cls + c] * expf(logit_scale[0]);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (int m = 0; m < M; ++m) {
            const size_t row = (size_t)b * L + 1 + m;
            acc += Tp[(((size_t)b * n_cls + c) * M + m) * N + n] * rnorm[row] * (float)f[row * D + d];
        }
        part[((size_t)b * n_cls * N + pc) * D + d] = gs * acc;
    }
}

bool args_ok(int B, int L, int D, int n_cls, int N, int mode) {
    return B > 0 && L > 1 && L - 1 <= OT_MAXM && D > 0 && (D & 3) == 0 && D <= 4 * 64 * OT_MAXV && n_cls > 0 &&
           n_cls <= OT_MAXC && N > 0 && N <= OT_MAXN && (mode == 1 || mode == 2);
}

}  // namespace

extern "C" int ffm_ot_head_fwd(const void* f, const float* tn, const float* logit_scale, float* rnorm, float* sim, float* T,
                               float* errs, int32_t* istop, float* tsum, float* logits_img, int B, int L, int D, int n_cls,
                               int N, int mode, float eps, float thresh, int max_iter, float top_percent, int dtype,
                               void* stream) {
    if (!f || !tn || !logit_scale || !rnorm || !sim || !T || !errs || !istop || !tsum || !logits_img) return FFM_EINVAL;
    if (!args_ok(B, L, D, n_cls, N, mode) || max_iter <= 0 || !(eps > 0.f)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int M = L - 1, nprob = B * n_cls, threads = ((M + 63) / 64) * 64;
    const int rows = B * L;
    // (:724-726) top_percent = min(torch.sum(xx).item(), TOP_PERCENT); xx sums to 1 per problem, so sum(xx) = #problems
    top_percent = top_percent < (float)nprob ? top_percent : (float)nprob;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((ot_sim_kernel<bf16_t>), dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_t*)f, tn, rnorm, sim, B, L, D, n_cls, N);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((ot_sim_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)f, tn, rnorm, sim, B, L, D, n_cls, N);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    if (mode == 1) {
        hipLaunchKernelGGL((ot_plan_kernel<1, 1>), dim3(nprob), dim3(threads), 0, s, sim, errs, istop, T, tsum, M, N, nprob, eps, top_percent, max_iter);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(ot_stop_kernel, dim3(1), dim3(256), 0, s, errs, istop, nprob, max_iter, (float)nprob * (float)M, thresh);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL((ot_plan_kernel<1, 2>), dim3(nprob), dim3(threads), 0, s, sim, errs, istop, T, tsum, M, N, nprob, eps, top_percent, max_iter);
    } else {
        hipLaunchKernelGGL((ot_plan_kernel<2, 1>), dim3(nprob), dim3(threads), 0, s, sim, errs, istop, T, tsum, M, N, nprob, eps, top_percent, max_iter);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(ot_stop_kernel, dim3(1), dim3(256), 0, s, errs, istop, nprob, max_iter, (float)nprob * (float)N, thresh);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL((ot_plan_kernel<2, 2>), dim3(nprob), dim3(threads), 0, s, sim, errs, istop, T, tsum, M, N, nprob, eps, top_percent, max_iter);
    }
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(ot_logits_kernel, dim3((nprob + 255) / 256), dim3(256), 0, s, tsum, logit_scale, logits_img, nprob);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_ot_head_bwd(const void* f, const float* tn, const float* logit_scale, const float* rnorm, const float* T,
                               const float* dlogits_img, void* df, float* dtn_part, int B, int L, int D, int n_cls, int N,
                               int dtype, void* stream) {
    if (!f || !tn || !logit_scale || !rnorm || !T || !dlogits_img || !df || !dtn_part) return FFM_EINVAL;
    if (!args_ok(B, L, D, n_cls, N, 1)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * L;
    if (dtype == FFM_BF16) {
        hipLaunchKernelGGL((ot_bwd_feat_kernel<bf16_t>), dim3((rows + 3) / 4), dim3(256), 0, s, (const bf16_t*)f, tn, logit_scale, rnorm, T, dlogits_img, (bf16_t*)df, B, L, D, n_cls, N);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL((ot_bwd_text_kernel<bf16_t>), dim3(B, n_cls * N), dim3(256), 0, s, (const bf16_t*)f, logit_scale, rnorm, T, dlogits_img, dtn_part, L, D, n_cls, N);
    } else if (dtype == FFM_F32) {
        hipLaunchKernelGGL((ot_bwd_feat_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)f, tn, logit_scale, rnorm, T, dlogits_img, (float*)df, B, L, D, n_cls, N);
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL((ot_bwd_text_kernel<float>), dim3(B, n_cls * N), dim3(256), 0, s, (const float*)f, logit_scale, rnorm, T, dlogits_img, dtn_part, L, D, n_cls, N);
    } else {
        return FFM_EINVAL;
    }
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
