// HBM-bound row kernels: LayerNorm fwd/bwd, patch gather + input normalisation,
// token assembly + ln_pre.  One wave per row, 4 elements per lane per step
// (8 B bf16 / 16 B f32), statistics in fp32 by wave shuffles.
#include "common.h"

namespace {

constexpr int MAXC = 4;   // width <= 8 * 64 * MAXC = 2048

// 8 consecutive fp32 parameters
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// MC: 8-element chunks per lane (width <= 8 * 64 * MC).  ViT widths (<= 1024) take MC = 2: half the registers of the
// general MC = 4 form (backward: 74 instead of 126).  It does not make the kernels faster - they are one wave
// generation of dependent latency (load, two wave reductions, store), not occupancy-bound: 10.4 -> 10.4 us measured,
// and forcing 7 waves per SIMD made it 11.1.
template <typename T, int MC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = width >> 3;
    const T* xr = x + (size_t)row * width;
    float v[MC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            Vec8<T>::load(xr + c * 8, v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float var = wave_sum(q) / (float)width;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    T* yr = y + (size_t)row * width;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float g[8], b[8], o[8];
            load8f(gamma + c * 8, g);
            load8f(beta + c * 8, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            Vec8<T>::store(yr + c * 8, o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// dx = rstd * (gy - mean(gy) - xhat * mean(gy * xhat)),  gy = gamma * dy
template <typename T, int MC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const T* __restrict__ res, T* __restrict__ out,
                                                            int rows, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = width >> 3;
    const float mean = mean_in[row], rstd = rstd_in[row];
    const size_t base = (size_t)row * width;
    float gy[MC][8], xh[MC][8], rr[MC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float d[8], xv[8], g[8];
            Vec8<T>::load(dy + base + c * 8, d);
            Vec8<T>::load(x + base + c * 8, xv);
            if (res) Vec8<T>::load(res + base + c * 8, rr[i]);
            load8f(gamma + c * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gy[i][e] = d[e] * g[e];
                xh[i][e] = (xv[e] - mean) * rstd;
                s1 += gy[i][e];
                s2 += gy[i][e] * xh[i][e];
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)width;
    const float m2 = wave_sum(s2) / (float)width;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = rstd * (gy[i][e] - m1 - xh[i][e] * m2);
                if (res) o[e] += rr[i][e];
            }
            Vec8<T>::store(out + base + c * 8, o);
        }
    }
}

// one thread per 4 consecutive kx of one (b, patch, c, ky)
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ cols, int B,
                                                       int H, int W, int ps, f32x4 mean3, f32x4 std3, int prenorm) {
    const int gw = W / ps, gh = H / ps;
    const int kdim = 3 * ps * ps;
    const int q4 = ps >> 2;                               // 4-element groups per patch row
    const size_t total = (size_t)B * gh * gw * 3 * ps * q4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int kx4 = (int)(t % q4); t /= q4;
        const int ky = (int)(t % ps); t /= ps;
        const int c = (int)(t % 3); t /= 3;
        const int px = (int)(t % gw); t /= gw;
        const int py = (int)(t % gh); t /= gh;
        const int b = (int)t;
        const float* src = img + (((size_t)b * 3 + c) * H + (py * ps + ky)) * W + px * ps + kx4 * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src);
        if (!prenorm) {
            // same op order as the reference: x/255, then -mean, then /std
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((v[e] / 255.0f) - mean3[c]) / std3[c];
        }
        T* dst = cols + ((size_t)(b * gh + py) * gw + px) * kdim + c * ps * ps + ky * ps + kx4 * 4;
        Vec4<T>::store(dst, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_lnpre_kernel(const T* __restrict__ patch, const T* __restrict__ cls,
                                                          const T* __restrict__ pos, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ x,
                                                          float* __restrict__ rowstat, int B, int L, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row % L;
    const int nchunk = width >> 3;                   // 8-element chunks: 16-byte loads in bf16 (width % 8 == 0: bad_width)
    const T* src = (l == 0) ? cls : patch + ((size_t)b * (L - 1) + (l - 1)) * width;
    const T* pr = pos + (size_t)l * width;
    constexpr int MAXC = 4;                          // width <= 8 * 64 * 4
    float v[MAXC][8], g[MAXC][8], bb[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float a[8], p[8];
            Vec8<T>::load(src + c * 8, a);
            Vec8<T>::load(pr + c * 8, p);
            load8f(gamma + c * 8, g[i]);             // (requested with the row: not a second round trip behind the sums)
            load8f(beta + c * 8, bb[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // the reference rounds (token + pos) to the activation dtype before ln_pre
                v[i][e] = Elem<T>::to_f(Elem<T>::from_f(a[e] + p[e]));
                s += v[i][e];
            }
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + 1e-5f);
    T* xr = x + (size_t)row * width;
    float so = 0.f, qo = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (v[i][e] - mean) * rstd * g[i][e] + bb[i][e];
                const float st = Elem<T>::to_f(Elem<T>::from_f(o[e]));       // the row as stored
                so += st;
                qo += st * st;
            }
            Vec8<T>::store(xr + c * 8, o);
        }
    }
    if (rowstat) {          // {sum, sum of squares} of the OUTPUT row: the first block's ln_1 folded into its qkv GEMM
        so = wave_sum(so);
        qo = wave_sum(qo);
        if (lane == 0) { rowstat[2 * (size_t)row] = so; rowstat[2 * (size_t)row + 1] = qo; }
    }
}

inline bool bad_width(int width) { return width <= 0 || (width & 7) || width > 8 * 64 * MAXC; }

}  // namespace

extern "C" int ffm_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean,
                                 float* rstd, int rows, int width, int dtype, void* stream) {
    if (!x || !y || !gamma || !beta || rows <= 0 || bad_width(width)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const bool narrow = width <= 8 * 64 * 2;
#define LN_FWD(T, MC) hipLaunchKernelGGL((layernorm_fwd_kernel<T, MC>), grid, block, 0, s, (const T*)x, (T*)y, gamma, beta, mean, rstd, rows, width)
    if (dtype == FFM_BF16) {
        if (narrow) LN_FWD(bf16_t, 2); else LN_FWD(bf16_t, 4);
    } else if (dtype == FFM_F32) {
        if (narrow) LN_FWD(float, 2); else LN_FWD(float, 4);
    } else {
        return FFM_EINVAL;
    }
#undef LN_FWD
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                 const float* rstd, const void* res, void* out, int rows, int width, int dtype,
                                 void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !out || rows <= 0 || bad_width(width)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    const bool narrow = width <= 8 * 64 * 2;
#define LN_BWD(T, MC) hipLaunchKernelGGL((layernorm_bwd_kernel<T, MC>), grid, block, 0, s, (const T*)dy, (const T*)x, gamma, mean, rstd, (const T*)res, (T*)out, rows, width)
    if (dtype == FFM_BF16) {
        if (narrow) LN_BWD(bf16_t, 2); else LN_BWD(bf16_t, 4);
    } else if (dtype == FFM_F32) {
        if (narrow) LN_BWD(float, 2); else LN_BWD(float, 4);
    } else {
        return FFM_EINVAL;
    }
#undef LN_BWD
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_patchify(const float* img, void* cols, int B, int H, int W, int patch, const float* mean3,
                            const float* std3, int prenormalised, int dtype, void* stream) {
    if (!img || !cols || B <= 0 || patch <= 0 || (patch & 3) || H % patch || W % patch) return FFM_EINVAL;
    if (!prenormalised && (!mean3 || !std3)) return FFM_EINVAL;
    // mean/std are HOST pointers to 3 floats (tiny, passed by value to the kernel)
    f32x4 m = {0.f, 0.f, 0.f, 0.f}, sd = {1.f, 1.f, 1.f, 1.f};
    if (!prenormalised) { m = (f32x4){mean3[0], mean3[1], mean3[2], 0.f}; sd = (f32x4){std3[0], std3[1], std3[2], 1.f}; }
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)B * (H / patch) * (W / patch) * 3 * patch * (patch / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, img, (bf16_t*)cols, B, H, W,
                           patch, m, sd, prenormalised);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((patchify_kernel<float>), dim3(blocks), dim3(256), 0, s, img, (float*)cols, B, H, W, patch,
                           m, sd, prenormalised);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_embed_lnpre(const void* patch, const void* cls, const void* pos, const float* gamma,
                               const float* beta, void* x, float* rowstat, int B, int L, int width, int dtype, void* stream) {
    if (!patch || !cls || !pos || !gamma || !beta || !x || B <= 0 || L <= 1 || bad_width(width)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((B * L + 3) / 4), block(256);
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((embed_lnpre_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)patch, (const bf16_t*)cls,
                           (const bf16_t*)pos, gamma, beta, (bf16_t*)x, rowstat, B, L, width);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((embed_lnpre_kernel<float>), grid, block, 0, s, (const float*)patch, (const float*)cls,
                           (const float*)pos, gamma, beta, (float*)x, rowstat, B, L, width);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
