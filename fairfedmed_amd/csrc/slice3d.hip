// 3D OCT input path (BASELINE.json configs[3]; trainers/GLP_OT_SVLoRA.py:585-595, 681-693):
//   x = image/255 viewed as [N = B*S, D, H, W];  c = conv5x5(x; W [3,D,5,5], b [3], pad 2)   (trainable)
//   y = (c - min_n) / (max_n - min_n + 1e-5)  per ViT image n over (3,H,W);  z = (y - mean)/std -> patches
// and its backward (conv weight / bias gradients; the input is data, so no dX of the conv).
// All of it is fp32 VALU work on ~1.6 MB of input per ViT image: HBM-bound.
#include "common.h"

namespace {

constexpr int TS = 16;          // output tile edge
constexpr int HALO = 2;
constexpr int IT = TS + 2 * HALO;
constexpr int MAXD = 16;        // slices per group (reference default 8)

__device__ __forceinline__ void stage_input(const float* __restrict__ img, float* in_s, int n, int D, int H, int W,
                                            int y0, int x0, int tid) {
    for (int idx = tid; idx < D * IT * IT; idx += 256) {
        const int c = idx / (IT * IT), rem = idx % (IT * IT);
        const int yy = y0 + rem / IT - HALO, xx = x0 + rem % IT - HALO;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[(((size_t)n * D + c) * H + yy) * W + xx] / 255.0f;
        in_s[idx] = v;
    }
}

// conv forward + per-block min/max
__global__ __launch_bounds__(256) void slice_conv_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             float* __restrict__ mm_part, int D, int H, int W) {
    __shared__ float in_s[MAXD * IT * IT];
    __shared__ float w_s[3 * MAXD * 25 + 3];
    __shared__ float red[2][256];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
    stage_input(img, in_s, n, D, H, W, y0, x0, tid);
    for (int i = tid; i < 3 * D * 25 + 3; i += 256) w_s[i] = i < 3 * D * 25 ? w[i] : bias[i - 3 * D * 25];
    __syncthreads();
    const int ty = tid / TS, tx = tid % TS;
    float acc[3] = {w_s[3 * D * 25], w_s[3 * D * 25 + 1], w_s[3 * D * 25 + 2]};
    for (int c = 0; c < D; ++c)
#pragma unroll
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float v = in_s[(c * IT + ty + ky) * IT + tx + kx];
#pragma unroll
                for (int o = 0; o < 3; ++o) acc[o] += v * w_s[((o * D + c) * 5 + ky) * 5 + kx];
            }
    const int y = y0 + ty, x = x0 + tx;
    float mn = INFINITY, mx = -INFINITY;
    if (y < H && x < W) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            out[(((size_t)n * 3 + o) * H + y) * W + x] = acc[o];
            mn = fminf(mn, acc[o]);
            mx = fmaxf(mx, acc[o]);
        }
    }
    red[0][tid] = mn;
    red[1][tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] = fminf(red[0][tid], red[0][tid + s]);
            red[1][tid] = fmaxf(red[1][tid], red[1][tid + s]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
        mm_part[((size_t)n * nblk + blk) * 2] = red[0][0];
        mm_part[((size_t)n * nblk + blk) * 2 + 1] = red[1][0];
    }
}

// mnmx[n] = {min, max}; counters of tied extrema are zeroed for the patchify pass
__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float* __restrict__ mm_part, float* __restrict__ mnmx,
                                                            int* __restrict__ cnt, int nblk) {
    __shared__ float red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < nblk; i += 256) {
        mn = fminf(mn, mm_part[((size_t)n * nblk + i) * 2]);
        mx = fmaxf(mx, mm_part[((size_t)n * nblk + i) * 2 + 1]);
    }
    red[0][tid] = mn;
    red[1][tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] = fminf(red[0][tid], red[0][tid + s]);
            red[1][tid] = fmaxf(red[1][tid], red[1][tid + s]);
        }
        __syncthreads();
    }
    if (tid == 0) { mnmx[n * 2] = red[0][0]; mnmx[n * 2 + 1] = red[1][0]; cnt[n * 2] = 0; cnt[n * 2 + 1] = 0; }
}

// cols[n*P + p][c*ps*ps + ky*ps + kx] = ((conv - mn)/(mx - mn + 1e-5) - mean[c]) / std[c]; counts the extrema
template <typename T>
__global__ __launch_bounds__(256) void patchify_minmax_kernel(const float* __restrict__ conv,
                                                              const float* __restrict__ mnmx, int* __restrict__ cnt,
                                                              T* __restrict__ cols, int N, int H, int W, int ps,
                                                              f32x4 mean3, f32x4 std3) {
    const int gw = W / ps, gh = H / ps, kdim = 3 * ps * ps, q4 = ps >> 2;
    const size_t total = (size_t)N * gh * gw * 3 * ps * q4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int kx4 = (int)(t % q4); t /= q4;
        const int ky = (int)(t % ps); t /= ps;
        const int c = (int)(t % 3); t /= 3;
        const int px = (int)(t % gw); t /= gw;
        const int py = (int)(t % gh); t /= gh;
        const int n = (int)t;
        const float mn = mnmx[n * 2], mx = mnmx[n * 2 + 1];
        const float d = mx - mn + 1e-5f;
        f32x4 v = *reinterpret_cast<const f32x4*>(conv + (((size_t)n * 3 + c) * H + (py * ps + ky)) * W + px * ps + kx4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (v[e] == mn) atomicAdd(&cnt[n * 2], 1);
            if (v[e] == mx) atomicAdd(&cnt[n * 2 + 1], 1);
            v[e] = (((v[e] - mn) / d) - mean3[c]) / std3[c];
        }
        Vec4<T>::store(cols + ((size_t)(n * gh + py) * gw + px) * kdim + c * ps * ps + ky * ps + kx4 * 4, v);
    }
}

// dcols -> gradient w.r.t. the conv output THROUGH THE DIRECT TERM ONLY (dy / d), plus per-block partial sums
// A = sum dy, Bs = sum dy * (c - mn) needed for the gradients that flow through min and max.
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_bwd_kernel(const T* __restrict__ dcols, const float* __restrict__ conv,
                                                             const float* __restrict__ mnmx, float* __restrict__ dconv,
                                                             float* __restrict__ ab_part, int N, int H, int W, int ps,
                                                             f32x4 std3) {
    __shared__ float red[2][256];
    const int gw = W / ps, gh = H / ps, kdim = 3 * ps * ps, q4 = ps >> 2;
    const int n = blockIdx.y;
    const int per_img = gh * gw * 3 * ps * q4;
    const float mn = mnmx[n * 2], mx = mnmx[n * 2 + 1];
    const float d = mx - mn + 1e-5f;
    float A = 0.f, Bs = 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < per_img; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int kx4 = t % q4; t /= q4;
        const int ky = t % ps; t /= ps;
        const int c = t % 3; t /= 3;
        const int px = t % gw; t /= gw;
        const int py = t;
        const f32x4 g = Vec4<T>::load(dcols + ((size_t)(n * gh + py) * gw + px) * kdim + c * ps * ps + ky * ps + kx4 * 4);
        const size_t off = (((size_t)n * 3 + c) * H + (py * ps + ky)) * W + px * ps + kx4 * 4;
        const f32x4 cv = *reinterpret_cast<const f32x4*>(conv + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = g[e] / std3[c];
            o[e] = dy / d;
            A += dy;
            Bs += dy * (cv[e] - mn);
        }
        *reinterpret_cast<f32x4*>(dconv + off) = o;
    }
    red[0][threadIdx.x] = A;
    red[1][threadIdx.x] = Bs;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ab_part[((size_t)n * gridDim.x + blockIdx.x) * 2] = red[0][0];
        ab_part[((size_t)n * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
    }
}

// y = (c - mn)/d, d = mx - mn + eps:  dL/dmn = -A/d + Bs/d^2,  dL/dmx = -Bs/d^2; torch's amin/amax backward
// spreads them evenly over tied extrema.  gmm[n] = {dmn / count_min, dmx / count_max}
__global__ void minmax_grad_kernel(const float* __restrict__ ab_part, const float* __restrict__ mnmx,
                                   const int* __restrict__ cnt, float* __restrict__ gmm, int N, int nblk) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float A = 0.f, Bs = 0.f;
    for (int i = 0; i < nblk; ++i) { A += ab_part[((size_t)n * nblk + i) * 2]; Bs += ab_part[((size_t)n * nblk + i) * 2 + 1]; }
    const float d = mnmx[n * 2 + 1] - mnmx[n * 2] + 1e-5f;
    gmm[n * 2] = (-A / d + Bs / (d * d)) / (float)max(cnt[n * 2], 1);
    gmm[n * 2 + 1] = (-Bs / (d * d)) / (float)max(cnt[n * 2 + 1], 1);
}

// conv weight / bias gradient partials: part[blk][o*D*25 + c*25 + ky*5 + kx] (+ 3 bias sums at the end)
__global__ __launch_bounds__(256) void slice_conv_wgrad_kernel(const float* __restrict__ img,
                                                               const float* __restrict__ conv,
                                                               const float* __restrict__ dconv,
                                                               const float* __restrict__ mnmx,
                                                               const float* __restrict__ gmm, float* __restrict__ part,
                                                               int D, int H, int W) {
    __shared__ float in_s[MAXD * IT * IT];
    __shared__ float g_s[3][TS * TS];
    const int tid = threadIdx.x, n = blockIdx.z;
    const int y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
    stage_input(img, in_s, n, D, H, W, y0, x0, tid);
    {
        const int ty = tid / TS, tx = tid % TS, y = y0 + ty, x = x0 + tx;
        const float mn = mnmx[n * 2], mx = mnmx[n * 2 + 1];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float g = 0.f;
            if (y < H && x < W) {
                const size_t off = (((size_t)n * 3 + o) * H + y) * W + x;
                const float cv = conv[off];
                g = dconv[off] + (cv == mn ? gmm[n * 2] : 0.f) + (cv == mx ? gmm[n * 2 + 1] : 0.f);
            }
            g_s[o][tid] = g;
        }
    }
    __syncthreads();
    const int nw = 3 * D * 25;
    const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int widx = tid; widx < nw + 3; widx += 256) {
        float s = 0.f;
        if (widx < nw) {
            const int o = widx / (D * 25), rem = widx % (D * 25);
            const int c = rem / 25, ky = (rem % 25) / 5, kx = rem % 5;
            for (int ty = 0; ty < TS; ++ty)
#pragma unroll
                for (int tx = 0; tx < TS; ++tx) s += g_s[o][ty * TS + tx] * in_s[(c * IT + ty + ky) * IT + tx + kx];
        } else {
            const int o = widx - nw;
            for (int i = 0; i < TS * TS; ++i) s += g_s[o][i];
        }
        part[(size_t)blk * (nw + 3) + widx] = s;
    }
}

// token assembly backward: LayerNorm backward of ln_pre, rows l >= 1 -> dpatch[b*P + l-1]
template <typename T>
__global__ __launch_bounds__(256) void embed_lnpre_bwd_kernel(const T* __restrict__ dx, const T* __restrict__ patch,
                                                              const T* __restrict__ pos, const float* __restrict__ gamma,
                                                              T* __restrict__ dpatch, int B, int L, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row % L;
    if (l == 0) return;                                  // the class token is a frozen parameter
    constexpr int MAXC = 8;
    const int nchunk = width >> 2;
    const T* src = patch + ((size_t)b * (L - 1) + (l - 1)) * width;
    const T* pr = pos + (size_t)l * width;
    f32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const f32x4 a = Vec4<T>::load(src + c * 4);
            const f32x4 p = Vec4<T>::load(pr + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[i][e] = Elem<T>::to_f(Elem<T>::from_f(a[e] + p[e])); s += v[i][e]; }
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + 1e-5f);
    f32x4 gy[MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const f32x4 d = Vec4<T>::load(dx + (size_t)row * width + c * 4);
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gy[i][e] = d[e] * g[e];
                v[i][e] = (v[i][e] - mean) * rstd;
                s1 += gy[i][e];
                s2 += gy[i][e] * v[i][e];
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)width, m2 = wave_sum(s2) / (float)width;
    T* dst = dpatch + ((size_t)b * (L - 1) + (l - 1)) * width;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (gy[i][e] - m1 - v[i][e] * m2);
            Vec4<T>::store(dst + c * 4, o);
        }
    }
}

}  // namespace

extern "C" int ffm_slice_blocks(int H, int W) { return ((H + TS - 1) / TS) * ((W + TS - 1) / TS); }

extern "C" int ffm_slice_conv_fwd(const float* img, const float* w, const float* bias, float* conv, float* mm_part,
                                  float* mnmx, int32_t* cnt, int N, int D, int H, int W, void* stream) {
    if (!img || !w || !bias || !conv || !mm_part || !mnmx || !cnt || N <= 0 || D <= 0 || D > MAXD || H <= 0 || W <= 0)
        return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, N);
    hipLaunchKernelGGL(slice_conv_fwd_kernel, grid, dim3(256), 0, s, img, w, bias, conv, mm_part, D, H, W);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(minmax_reduce_kernel, dim3(N), dim3(256), 0, s, mm_part, mnmx, cnt, (int)(grid.x * grid.y));
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_patchify_minmax(const float* conv, const float* mnmx, int32_t* cnt, void* cols, int N, int H, int W,
                                   int patch, const float* mean3, const float* std3, int dtype, void* stream) {
    if (!conv || !mnmx || !cnt || !cols || !mean3 || !std3 || N <= 0 || patch <= 0 || (patch & 3) || H % patch || W % patch)
        return FFM_EINVAL;
    const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
    const size_t total = (size_t)N * (H / patch) * (W / patch) * 3 * patch * (patch / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((patchify_minmax_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, conv, mnmx, cnt, (bf16_t*)cols,
                           N, H, W, patch, m, sd);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((patchify_minmax_kernel<float>), dim3(blocks), dim3(256), 0, s, conv, mnmx, cnt, (float*)cols, N,
                           H, W, patch, m, sd);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_embed_lnpre_bwd(const void* dx, const void* patch, const void* pos, const float* gamma, void* dpatch,
                                   int B, int L, int width, int dtype, void* stream) {
    if (!dx || !patch || !pos || !gamma || !dpatch || B <= 0 || L <= 1 || width <= 0 || (width & 3) || width > 2048)
        return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((B * L + 3) / 4), block(256);
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((embed_lnpre_bwd_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dx, (const bf16_t*)patch,
                           (const bf16_t*)pos, gamma, (bf16_t*)dpatch, B, L, width);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((embed_lnpre_bwd_kernel<float>), grid, block, 0, s, (const float*)dx, (const float*)patch,
                           (const float*)pos, gamma, (float*)dpatch, B, L, width);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

#define SLICE_BWD_BLOCKS 64   /* partial sums per ViT image in the un-patchify pass */

extern "C" int ffm_slice_bwd(const void* dcols, const float* img, const float* conv, const float* mnmx, const int32_t* cnt,
                             float* dconv, float* ab_part, float* gmm, float* wpart, int N, int D, int H, int W,
                             int patch, const float* std3, int dtype, void* stream) {
    if (!dcols || !img || !conv || !mnmx || !cnt || !dconv || !ab_part || !gmm || !wpart || !std3) return FFM_EINVAL;
    if (N <= 0 || D <= 0 || D > MAXD || patch <= 0 || (patch & 3) || H % patch || W % patch) return FFM_EINVAL;
    const f32x4 sd = {std3[0], std3[1], std3[2], 1.f};
    hipStream_t s = (hipStream_t)stream;
    dim3 g1(SLICE_BWD_BLOCKS, N);
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((unpatchify_bwd_kernel<bf16_t>), g1, dim3(256), 0, s, (const bf16_t*)dcols, conv, mnmx, dconv,
                           ab_part, N, H, W, patch, sd);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((unpatchify_bwd_kernel<float>), g1, dim3(256), 0, s, (const float*)dcols, conv, mnmx, dconv,
                           ab_part, N, H, W, patch, sd);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(minmax_grad_kernel, dim3((N + 63) / 64), dim3(64), 0, s, ab_part, mnmx, cnt, gmm, N,
                       SLICE_BWD_BLOCKS);
    FFM_CHECK_LAUNCH();
    dim3 g3((W + TS - 1) / TS, (H + TS - 1) / TS, N);
    hipLaunchKernelGGL(slice_conv_wgrad_kernel, g3, dim3(256), 0, s, img, conv, dconv, mnmx, gmm, wpart, D, H, W);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_slice_bwd_ab_blocks(void) { return SLICE_BWD_BLOCKS; }
