// Logits head with OT='None' (trainers/GLP_OT_SVLoRA.py:713-757), the
// cross-entropy loss (:908) and their backward.  One block of 16 waves per ViT image (the backward also splits the
// tokens over 4 blocks): with 4 waves a wave walked 49 tokens one after the other, 51 us for 6.5 MB.
#include "common.h"

namespace {

constexpr int HD_MAXV = 4;     // D <= 4 * 64 * HD_MAXV = 1024 (512 ViT-B/16, 1024 RN50)
constexpr int HD_MAXC = 8;     // n_cls <= 8

// fbar[b] = mean_{l>=1} f[b,l]/|f[b,l]|;  logits[b][c] = e^ls <fbar[b], tbar[c]>
constexpr int HD_NW = 16;      // waves per block

template <typename T>
__global__ __launch_bounds__(64 * HD_NW) void head_fwd_kernel(const T* __restrict__ f, const float* __restrict__ tbar,
                                                       const float* __restrict__ logit_scale,
                                                       float* __restrict__ fbar, float* __restrict__ rnorm,
                                                       float* __restrict__ logits, int L, int D, int n_cls) {
    __shared__ float red[HD_NW / 2][4 * 64 * HD_MAXV];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D >> 2;
    f32x4 acc[HD_MAXV];
#pragma unroll
    for (int i = 0; i < HD_MAXV; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (four rows in flight per wave instead of one: measured, 21.4 against 20.7 us - this kernel does not wait for its loads)
    for (int l = 1 + wave; l < L; l += HD_NW) {
        const T* fr = f + ((size_t)b * L + l) * D;
        f32x4 v[HD_MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                v[i] = Vec4<T>::load(fr + c * 4);
                ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
        }
        const float nrm = sqrtf(wave_sum(ss));
        const float rn = 1.0f / fmaxf(nrm, 1e-12f);          // F.normalize eps
        if (lane == 0) rnorm[(size_t)b * L + l] = rn;
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] += v[i][e] * rn;
            }
        }
    }
    if (wave == 0 && lane == 0) rnorm[(size_t)b * L] = 0.f;
    // deterministic two-step combine: the upper half of the waves parks its sums, the lower half adds and parks
    if (wave >= HD_NW / 2) {
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave - HD_NW / 2][c * 4 + e] = acc[i][e];
            }
        }
    }
    __syncthreads();
    if (wave < HD_NW / 2) {
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][c * 4 + e] += acc[i][e];
            }
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)(L - 1);
    for (int d = threadIdx.x; d < D; d += 64 * HD_NW) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < HD_NW / 2; ++w) s += red[w][d];
        s *= inv;
        red[0][d] = s;
        fbar[(size_t)b * D + d] = s;
    }
    __syncthreads();
    const float scale = expf(logit_scale[0]);
    for (int c = wave; c < n_cls; c += HD_NW) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += red[0][d] * tbar[(size_t)c * D + d];
        s = wave_sum(s);
        if (lane == 0) logits[(size_t)b * n_cls + c] = scale * s;
    }
}

// single block: slice mean, softmax CE, dlogits
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits_img,
                                                      const int64_t* __restrict__ label, float* __restrict__ logits,
                                                      float* __restrict__ prob, float* __restrict__ loss,
                                                      float* __restrict__ dlogits_img, int32_t* __restrict__ finite,
                                                      int nb, int S, int n_cls) {
    __shared__ float part[256];
    float my = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) {
        float lg[HD_MAXC];
        float mx = -INFINITY;
        for (int c = 0; c < n_cls; ++c) {
            float s = 0.f;
            for (int k = 0; k < S; ++k) s += logits_img[((size_t)b * S + k) * n_cls + c];
            lg[c] = s / (float)S;
            logits[(size_t)b * n_cls + c] = lg[c];
            mx = fmaxf(mx, lg[c]);
        }
        float se = 0.f;
        for (int c = 0; c < n_cls; ++c) se += expf(lg[c] - mx);
        const float lse = mx + logf(se);
        const int y = (int)label[b];
        my += lse - lg[y];
        for (int c = 0; c < n_cls; ++c) {
            const float pc = expf(lg[c] - lse);
            prob[(size_t)b * n_cls + c] = pc;
            const float dl = (pc - (c == y ? 1.f : 0.f)) / (float)nb / (float)S;
            for (int k = 0; k < S; ++k) dlogits_img[((size_t)b * S + k) * n_cls + c] = dl;
        }
    }
    part[threadIdx.x] = my;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l = part[0] / (float)nb;
        loss[0] = l;
        if (finite) finite[0] = isfinite(l) ? 1 : 0;
    }
}

// dfbar[b] = e^ls * sum_c dlogits[b][c] tbar[c];  y = f*rn;  df = (dy - y <y,dy>) rn, dy = dfbar/(L-1)
template <typename T>
__global__ __launch_bounds__(64 * HD_NW) void head_bwd_kernel(const T* __restrict__ f, const float* __restrict__ tbar,
                                                       const float* __restrict__ logit_scale,
                                                       const float* __restrict__ rnorm,
                                                       const float* __restrict__ dlogits, T* __restrict__ df, int L,
                                                       int D, int n_cls) {
    __shared__ float dyb[4 * 64 * HD_MAXV];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = D >> 2;
    const float scale = expf(logit_scale[0]) / (float)(L - 1);
    for (int d = threadIdx.x; d < D; d += 64 * HD_NW) {
        float s = 0.f;
        for (int c = 0; c < n_cls; ++c) s += dlogits[(size_t)b * n_cls + c] * tbar[(size_t)c * D + d];
        dyb[d] = s * scale;
    }
    __syncthreads();
    for (int l = blockIdx.y * HD_NW + wave; l < L; l += HD_NW * gridDim.y) {
        T* dr = df + ((size_t)b * L + l) * D;
        if (l == 0) {
#pragma unroll
            for (int i = 0; i < HD_MAXV; ++i) {
                const int c = lane + 64 * i;
                if (c < nchunk) Vec4<T>::store(dr + c * 4, (f32x4){0.f, 0.f, 0.f, 0.f});
            }
            continue;
        }
        const T* fr = f + ((size_t)b * L + l) * D;
        const float rn = rnorm[(size_t)b * L + l];
        f32x4 y[HD_MAXV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                const f32x4 v = Vec4<T>::load(fr + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[i][e] = v[e] * rn; dot += y[i][e] * dyb[c * 4 + e]; }
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < HD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (dyb[c * 4 + e] - y[i][e] * dot) * rn;
                Vec4<T>::store(dr + c * 4, o);
            }
        }
    }
}

// dtbar[c][d] = e^ls * sum_b dlogits[b][c] * fbar[b][d]
__global__ __launch_bounds__(256) void head_dtbar_kernel(const float* __restrict__ fbar,
                                                         const float* __restrict__ logit_scale,
                                                         const float* __restrict__ dlogits, float* __restrict__ dtbar,
                                                         int B, int D, int n_cls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cls * D) return;
    const int c = i / D, d = i % D;
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {                             // eight images' loads in flight (a 32-deep chain: 10 us); same order
        float dl[8], fb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            dl[u] = dlogits[(size_t)(b + u) * n_cls + c];
            fb[u] = fbar[(size_t)(b + u) * D + d];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += dl[u] * fb[u];
    }
    for (; b < B; ++b) s += dlogits[(size_t)b * n_cls + c] * fbar[(size_t)b * D + d];
    dtbar[i] = s * expf(logit_scale[0]);
}

}  // namespace

extern "C" int ffm_head_fwd(const void* f, const float* tbar, const float* logit_scale, float* fbar, float* rnorm,
                            float* logits_img, int B, int L, int D, int n_cls, int dtype, void* stream) {
    if (!f || !tbar || !logit_scale || !fbar || !rnorm || !logits_img) return FFM_EINVAL;
    if (B <= 0 || L <= 1 || D <= 0 || (D & 3) || D > 4 * 64 * HD_MAXV || n_cls <= 0 || n_cls > HD_MAXC) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((head_fwd_kernel<bf16_t>), dim3(B), dim3(64 * HD_NW), 0, s, (const bf16_t*)f, tbar, logit_scale,
                           fbar, rnorm, logits_img, L, D, n_cls);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((head_fwd_kernel<float>), dim3(B), dim3(64 * HD_NW), 0, s, (const float*)f, tbar, logit_scale,
                           fbar, rnorm, logits_img, L, D, n_cls);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_ce_loss(const float* logits_img, const int64_t* label, float* logits, float* prob, float* loss,
                           float* dlogits_img, int32_t* finite_flag, int nb, int S, int n_cls, void* stream) {
    if (!logits_img || !label || !logits || !prob || !loss || !dlogits_img) return FFM_EINVAL;
    if (nb <= 0 || S <= 0 || n_cls <= 0 || n_cls > HD_MAXC) return FFM_EINVAL;
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits_img, label, logits, prob,
                       loss, dlogits_img, finite_flag, nb, S, n_cls);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_head_bwd(const void* f, const float* tbar, const float* logit_scale, const float* fbar,
                            const float* rnorm, const float* dlogits_img, void* df, float* dtbar, int B, int L, int D,
                            int n_cls, int dtype, void* stream) {
    if (!f || !tbar || !logit_scale || !fbar || !rnorm || !dlogits_img || !df || !dtbar) return FFM_EINVAL;
    if (B <= 0 || L <= 1 || D <= 0 || (D & 3) || D > 4 * 64 * HD_MAXV || n_cls <= 0 || n_cls > HD_MAXC) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((head_bwd_kernel<bf16_t>), dim3(B, 4), dim3(64 * HD_NW), 0, s, (const bf16_t*)f, tbar, logit_scale,
                           rnorm, dlogits_img, (bf16_t*)df, L, D, n_cls);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((head_bwd_kernel<float>), dim3(B, 4), dim3(64 * HD_NW), 0, s, (const float*)f, tbar, logit_scale,
                           rnorm, dlogits_img, (float*)df, L, D, n_cls);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_dtbar_kernel, dim3((n_cls * D + 255) / 256), dim3(256), 0, s, fbar, logit_scale,
                       dlogits_img, dtbar, B, D, n_cls);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
