// Flat-buffer parameter kernels: fused SGD-momentum over all trainable tensors,
// the FedAvg round-boundary helpers, and load-time dtype casts / transposes.
#include "common.h"

namespace {

// torch.optim.SGD, dampening 0, no nesterov (Dassl/dassl/optim/optimizer.py:105-113)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, int64_t n, float lr, float mu, float wd,
                                                  int first) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float d = g[i] + wd * pi;
        const float b = first ? d : mu * buf[i] + d;
        buf[i] = b;
        p[i] = pi - lr * b;
    }
}

// `repeats` applications of the same update on the SAME gradient in one pass over memory: the reference registers
// 'prompt_learner' and 'image_encoder' with ONE optimizer (trainers/GLP_OT_SVLoRA.py:866-870) and Dassl's model_update
// steps every registered name (Dassl/dassl/engine/trainer.py:333-337), so optim.step() runs twice per batch.
__global__ __launch_bounds__(256) void sgd_n_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ buf, int64_t n, float lr, float mu, float wd,
                                                    int first, int repeats) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i];
        const float gi = g[i];
        float b = first ? 0.f : buf[i];
        for (int k = 0; k < repeats; ++k) {
            const float d = gi + wd * pi;
            b = (first && k == 0) ? d : mu * b + d;
            pi = pi - lr * b;
        }
        buf[i] = b;
        p[i] = pi;
    }
}

// same update with the hyper-parameters read from device memory (hp = {lr, momentum, weight_decay}),
// so that a captured hipGraph keeps working when the LR scheduler changes lr.  With a zero-initialised
// momentum buffer the general formula equals torch's first-step rule (mu*0 + d == d exactly).
__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ buf, int64_t n,
                                                      const float* __restrict__ hp) {
    const float lr = hp[0], mu = hp[1], wd = hp[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float b = mu * buf[i] + (g[i] + wd * pi);
        buf[i] = b;
        p[i] = pi - lr * b;
    }
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ p, const float* __restrict__ w,
                                                       float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = p[i] * w[i];
}

// a rank that holds several clients of a round: acc += p (.) w, product and sum rounded separately (no FMA), as the
// reference's `w_avg[key] += w[idx][key] * weight` (utils/fed_utils.py:79-86) rounds them
__global__ __launch_bounds__(256) void scale_acc_kernel(const float* __restrict__ p, const float* __restrict__ w,
                                                        float* __restrict__ acc, int64_t n) {
#pragma clang fp contract(off)      // (hipcc contracts a * b + c into an FMA by default, and __fmul_rn is a plain product in HIP)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc[i] = acc[i] + p[i] * w[i];
}

// utils/fed_utils.py:88-98: optional shared_half_s, then EMA with the previous global
__global__ __launch_bounds__(256) void shared_half_kernel(float* __restrict__ avg, const int64_t* __restrict__ offs,
                                                          int n_s, int G, int r) {
    const int blk = blockIdx.x;
    if (blk >= n_s) return;
    float* s = avg + offs[blk];
    const int half = r / 2;
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        float m = 0.f;
        for (int g = 0; g < G; ++g) m += s[g * r + j];
        m /= (float)G;
        for (int g = 0; g < G; ++g) s[g * r + j] = m;
    }
}

// (prev and out may be the same buffer: the aggregator updates its global weights in place)
__global__ __launch_bounds__(256) void ema_kernel(const float* __restrict__ avg, const float* prev, float* out, int64_t n,
                                                  float beta) {
#pragma clang fp contract(off)      // three roundings, as `(1 - b) * avg + b * w_g` has in the reference (utils/fed_utils.py:98)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (1.0f - beta) * avg[i] + beta * prev[i];
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = Elem<T>::from_f(src[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst,
                                                          int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = Elem<T>::to_f(src[i]);
}

// dst[c][r] = src[r][c], 32x32 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                             int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * rows + r] = Elem<T>::from_f(tile[tx][i]);
    }
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// p *= s in place; *finite_flag = 0 if any product is not finite (the overflow guard of the IEEE-half mode's gradient scale)
__global__ __launch_bounds__(256) void scale_check_kernel(float* __restrict__ p, float sc, int64_t n, int32_t* __restrict__ finite_flag) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = p[i] * sc;
        p[i] = v;
        bad |= !(fabsf(v) <= 3.4028234e38f);
    }
    if (finite_flag && __any(bad) && (threadIdx.x & 63) == 0) *finite_flag = 0;
}

// IEEE-half mode, dynamic gradient scale held in DEVICE memory so that recorded launch plans / captured graphs stay valid
// while it changes: st = {scale, 1/scale, ok, good_run, overflows, max_scale, growth_interval, min_scale} (floats).
// A step is  loss_scale(dlogits) -> backward -> unscale_check(grad) -> sgd_gated -> scale_update.
__global__ __launch_bounds__(256) void loss_scale_kernel(float* __restrict__ p, int64_t n, float* __restrict__ st) {
    const float sc = st[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] *= sc;
    // (ok is read by kernels behind this one in stream order only; every block has read st[0] from a different word)
    if (blockIdx.x == 0 && threadIdx.x == 0) st[2] = 1.0f;
}

__global__ __launch_bounds__(256) void unscale_check_kernel(float* __restrict__ p, int64_t n, float* __restrict__ st) {
    const float inv = st[1];
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = p[i] * inv;
        p[i] = v;
        bad |= !(fabsf(v) <= 3.4028234e38f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) st[2] = 0.0f;
}

__global__ __launch_bounds__(256) void sgd_gated_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ buf, int64_t n, float lr, float mu, float wd,
                                                        int first, int repeats, const float* __restrict__ st) {
    if (st[2] == 0.0f) return;                    // the gradients overflowed: the step is skipped, momentum untouched
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i];
        const float gi = g[i];
        float b = first ? 0.f : buf[i];
        for (int k = 0; k < repeats; ++k) {
            const float d = gi + wd * pi;
            b = (first && k == 0) ? d : mu * b + d;
            pi = pi - lr * b;
        }
        buf[i] = b;
        p[i] = pi;
    }
}

__global__ void scale_update_kernel(float* __restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sc = st[0];
    if (st[2] == 0.0f) {                          // overflow: halve (torch.cuda.amp.GradScaler's backoff), count it
        sc = fmaxf(sc * 0.5f, st[7]);
        st[3] = 0.0f;
        st[4] += 1.0f;
    } else {
        st[3] += 1.0f;
        if (st[6] > 0.0f && st[3] >= st[6] && sc < st[5]) {
            sc = fminf(sc * 2.0f, st[5]);
            st[3] = 0.0f;
        }
    }
    st[0] = sc;
    st[1] = 1.0f / sc;
}

}  // namespace

extern "C" int ffm_abi_version(void) { return FFM_ABI_VERSION; }

extern "C" int ffm_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                float weight_decay, int first_step, void* stream) {
    if (!p || !g || !buf || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum,
                       weight_decay, first_step);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_sgd_momentum_n(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                  float weight_decay, int first_step, int repeats, void* stream) {
    if (!p || !g || !buf || n <= 0 || repeats < 1 || repeats > 16) return FFM_EINVAL;
    hipLaunchKernelGGL(sgd_n_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum,
                       weight_decay, first_step, repeats);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_sgd_momentum_dev(float* p, const float* g, float* buf, int64_t n, const float* hp, void* stream) {
    if (!p || !g || !buf || !hp || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(sgd_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, hp);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_scale_by(const float* p, const float* w, float* out, int64_t n, void* stream) {
    if (!p || !w || !out || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, w, out, n);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_scale_acc(const float* p, const float* w, float* acc, int64_t n, void* stream) {
    if (!p || !w || !acc || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(scale_acc_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, w, acc, n);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_scale_check(float* p, float scale, int64_t n, int32_t* finite_flag, void* stream) {
    if (!p || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(scale_check_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, scale, n, finite_flag);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_loss_scale(float* p, int64_t n, float* state, void* stream) {
    if (!p || !state || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(loss_scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, state);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_unscale_check(float* g, int64_t n, float* state, void* stream) {
    if (!g || !state || n <= 0) return FFM_EINVAL;
    hipLaunchKernelGGL(unscale_check_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, state);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_sgd_momentum_gated(float* p, const float* g, float* buf, int64_t n, float lr, float momentum,
                                      float weight_decay, int first_step, int repeats, float* state, void* stream) {
    if (!p || !g || !buf || !state || n <= 0 || repeats < 1 || repeats > 16) return FFM_EINVAL;
    hipLaunchKernelGGL(sgd_gated_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum,
                       weight_decay, first_step, repeats, state);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_fedavg_finish(float* avg, const float* prev, float* out, int64_t n, const int64_t* s_offsets,
                                 int n_s, int G, int r, int shared_half_s, float beta, void* stream) {
    if (!avg || !prev || !out || n <= 0) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (shared_half_s && n_s > 0) {
        if (!s_offsets || G <= 0 || r <= 0) return FFM_EINVAL;
        hipLaunchKernelGGL(shared_half_kernel, dim3(n_s), dim3(64), 0, s, avg, s_offsets, n_s, G, r);
        FFM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, s, avg, prev, out, n, beta);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_cast_f32_to(const float* src, void* dst, int64_t n, int dtype, void* stream) {
    if (!src || !dst || n <= 0) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((cast_from_f32_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, src, (bf16_t*)dst, n);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, src, (float*)dst, n);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream) {
    if (!src || !dst || n <= 0) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((cast_to_f32_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)src, dst, n);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((cast_to_f32_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (const float*)src, dst, n);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_transpose_cast(const float* src, void* dst, int rows, int cols, int dtype, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (dtype == FFM_BF16)
        hipLaunchKernelGGL((transpose_cast_kernel<bf16_t>), grid, dim3(256), 0, s, src, (bf16_t*)dst, rows, cols);
    else if (dtype == FFM_F32)
        hipLaunchKernelGGL((transpose_cast_kernel<float>), grid, dim3(256), 0, s, src, (float*)dst, rows, cols);
    else
        return FFM_EINVAL;
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
