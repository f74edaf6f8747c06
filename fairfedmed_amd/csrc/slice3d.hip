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

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

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


// ---------------------------------------------------------------------------------------------------------------------
// Second generation of the two convolution passes (round 5).  The first one staged a 16 x 16 tile with its halo in LDS
// and read every input AND every weight back from LDS per multiply (forward: 800 LDS reads for 600 FMAs per pixel;
// weight gradient: two LDS reads per FMA and 196 partial rows of 603 floats per image): 371 us and 552 + 141 us per
// step of BASELINE configs[3] (100 slice groups of 8 x 224 x 224) against ~90 us of fp32 FMA issue for either.
// Here a lane owns a COLUMN x of a 64-column strip and walks down the rows:
//   forward   the wave keeps FW_R output rows x 3 channels in registers; for every input channel and input row it loads
//             the five values x - 2 .. x + 2 once (coalesced dword loads, L1 hits for the overlap) and feeds the <= 5
//             output rows they touch - 75 FMAs per 5 loads, the weights of the channel in SGPRs (uniform s_loads);
//   gradient  a wave owns ONE input channel: 75 + 3 accumulators per lane (3 outputs x 5 x 5 taps), a rolling window
//             of five input rows in registers, one cross-lane reduction per wave at the end of its WG_R rows; the
//             partial rows shrink from 196 to 16 per image (the reduction behind them from 141 to ~10 us).
// Same arithmetic per element as before (image / 255 became image * (1 / 255): one ulp of the input at most).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FW_R = 14;        // output rows per wave, forward (224 = 16 x 14)
constexpr int WG_R = 56;        // output rows per wave, weight gradient (224 = 4 x 56)
constexpr int SW = 64;          // columns per strip = lanes

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void slice_conv_fwd2_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              float* __restrict__ mm_part, int D, int H, int W, int nrb) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z, rb = blockIdx.y * 4 + wave;
    if (rb >= nrb) return;
    const int x = blockIdx.x * SW + lane, y0 = rb * FW_R;
    int xc[5];
    bool xok[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
        const int xx = x + kx - 2;
        xok[kx] = xx >= 0 && xx < W;
        xc[kx] = xx < 0 ? 0 : (xx < W ? xx : W - 1);
    }
    float acc[FW_R][3];
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
#pragma unroll
    for (int r = 0; r < FW_R; ++r) { acc[r][0] = b0; acc[r][1] = b1; acc[r][2] = b2; }
    int xb[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) xb[kx] = xc[kx] * 4;
    for (int c = 0; c < D; ++c) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + ((size_t)n * D + c) * H * W), 0, H * W * 4, 0x00020000);
        float wr[3][25];                                            // uniform addresses: scalar loads, SGPR operands
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < 25; ++k) wr[o][k] = w[(o * D + c) * 25 + k];
        // input rows two ahead of the FMAs that use them; the scheduling barrier at the end of every row keeps hipcc from
        // hoisting all 90 loads of a channel to its top (240 registers, and spills at 128)
        float v[FW_R + 4][5];
        auto load_row = [&](int j) {
            const int yy = y0 - 2 + j;
            const int yc = yy < 0 ? 0 : (yy < H ? yy : H - 1);      // (a row outside the image: any row, zeroed at its use)
            // buffer loads: plane descriptor + uniform row offset (SGPR) + the lane's column offset - no 64-bit lane
            // addresses (flat loads made hipcc keep 90 loop-invariant address pairs alive: spills at 128 registers)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
                v[j][kx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, xb[kx], yc * W * 4, 0));
        };
        // the raw value is scaled / zeroed where it is USED: done at the load, the wait for it would sit there as well
        auto use_row = [&](int j) {
            const int yy = y0 - 2 + j;
            const bool rowok = yy >= 0 && yy < H;
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) v[j][kx] = (rowok && xok[kx]) ? v[j][kx] * (1.0f / 255.0f) : 0.f;
        };
        load_row(0);
        load_row(1);
#pragma unroll
        for (int j = 0; j < FW_R + 4; ++j) {
            if (j + 2 < FW_R + 4) load_row(j + 2);
            use_row(j);
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int r = j - ky;
                if (r < 0 || r >= FW_R) continue;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                    for (int o = 0; o < 3; ++o) acc[r][o] = fmaf(v[j][kx], wr[o][ky * 5 + kx], acc[r][o]);
            }
            asm volatile("" ::: "memory");                          // (the loads are readonly intrinsics: pins them in the IR too)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float mn = INFINITY, mx = -INFINITY;
    if (x < W) {
#pragma unroll
        for (int r = 0; r < FW_R; ++r) {
            const int y = y0 + r;
            if (y >= H) break;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                out[(((size_t)n * 3 + o) * H + y) * W + x] = acc[r][o];
                mn = fminf(mn, acc[r][o]);
                mx = fmaxf(mx, acc[r][o]);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) {
        const int nblk = gridDim.x * nrb, blk = blockIdx.x * nrb + rb;
        mm_part[((size_t)n * nblk + blk) * 2] = mn;
        mm_part[((size_t)n * nblk + blk) * 2 + 1] = mx;
    }
}

// part[(n * nblk + blk)][o*D*25 + c*25 + ky*5 + kx] (+ 3 bias sums at the end), blk = strip * nrb + row block; the wave of
// channel c writes that channel's 75 entries, the wave of channel 0 the three bias sums as well.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void slice_conv_wgrad2_kernel(const float* __restrict__ img, const float* __restrict__ conv,
                                                                const float* __restrict__ dconv,
                                                                const float* __restrict__ mnmx,
                                                                const float* __restrict__ gmm, float* __restrict__ part,
                                                                int D, int H, int W, int nrb, int cgroups) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z / cgroups, c = (blockIdx.z % cgroups) * 4 + wave;
    if (c >= D) return;
    const int rb = blockIdx.y, y0 = rb * WG_R;
    const int rows = (H - y0) < WG_R ? (H - y0) : WG_R;
    const int x = blockIdx.x * SW + lane;
    int xc[5];
    bool xok[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
        const int xx = x + kx - 2;
        xok[kx] = xx >= 0 && xx < W;
        xc[kx] = xx < 0 ? 0 : (xx < W ? xx : W - 1);
    }
    const bool mine = x < W;
    const int xs = mine ? x : W - 1;
    const float mn = mnmx[n * 2], mx = mnmx[n * 2 + 1], g0 = gmm[n * 2], g1 = gmm[n * 2 + 1];
    const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + ((size_t)n * D + c) * H * W), 0, H * W * 4, 0x00020000);
    // outputs 0 and 1 share a register pair per tap (v_pk_fma_f32 with the window value broadcast by op_sel), output 2 a
    // plain FMA: 50 instructions per row instead of 75
    f32x2 acc01[25];
    float acc2[25], bs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 25; ++k) { acc01[k] = (f32x2){0.f, 0.f}; acc2[k] = 0.f; }
    float win[5][5];                                                // input rows y - 2 .. y + 2 of the current output row
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) win[i][kx] = 0.f;
    // step t takes input row y0 - 2 + t into slot t % 5; from t = 4 on, output row r = t - 4 is complete: its tap row ky is
    // input row y0 - 2 + r + ky = slot (t + ky + 1) % 5.  Unrolled by 5 so that every slot index is a compile-time constant.
    // Everything a step reads from memory was requested one step earlier (raw values; scaled / masked at their use, so that
    // the waits sit there and not at the loads), and a barrier for the compiler at the end of a step keeps that order.
    const int steps = rows + 4;
    int xb[5];
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) xb[kx] = xc[kx] * 4;
    const auto cplane = __builtin_amdgcn_make_buffer_rsrc((void*)(conv + (size_t)n * 3 * H * W), 0, 3 * H * W * 4, 0x00020000);
    const auto dplane = __builtin_amdgcn_make_buffer_rsrc((void*)(dconv + (size_t)n * 3 * H * W), 0, 3 * H * W * 4, 0x00020000);
    float rawn[5], cvn[3], dcn[3];
    auto prefetch = [&](int t) {
        const int yy = y0 - 2 + t;
        const int yc = yy < 0 ? 0 : (yy < H ? yy : H - 1);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx)
            rawn[kx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, xb[kx], yc * W * 4, 0));
        const int y = y0 + t - 4, yo = y < 0 ? 0 : (y < H ? y : H - 1);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            cvn[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cplane, xs * 4, (o * H + yo) * W * 4, 0));
            dcn[o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dplane, xs * 4, (o * H + yo) * W * 4, 0));
        }
    };
    prefetch(0);
    for (int q = 0; q * 5 < steps; ++q) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = q * 5 + i;
            if (t >= steps) break;
            const int yy = y0 - 2 + t;
            const bool rowok = yy >= 0 && yy < H;                   // wave-uniform
            float cv[3], dc[3];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) win[i][kx] = (rowok && xok[kx]) ? rawn[kx] * (1.0f / 255.0f) : 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) { cv[o] = cvn[o]; dc[o] = dcn[o]; }
            if (t + 1 < steps) prefetch(t + 1);
            if (t >= 4) {
                float g[3];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float gv = dc[o] + (cv[o] == mn ? g0 : 0.f) + (cv[o] == mx ? g1 : 0.f);
                    g[o] = mine ? gv : 0.f;
                    bs[o] += g[o];
                }
                const f32x2 g01 = {g[0], g[1]};
#pragma unroll
                for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) {
                        const float wv = win[(i + ky + 1) % 5][kx];
                        acc01[ky * 5 + kx] = __builtin_elementwise_fma(g01, (f32x2){wv, wv}, acc01[ky * 5 + kx]);
                        acc2[ky * 5 + kx] = fmaf(g[2], wv, acc2[ky * 5 + kx]);
                    }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int nw = 3 * D * 25, nblk = gridDim.x * nrb, blk = blockIdx.x * nrb + rb;
    float* __restrict__ prow = part + ((size_t)n * nblk + blk) * (nw + 3);
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            const float s = wave_sum(o == 2 ? acc2[k] : acc01[k][o]);
            if (lane == 0) prow[(o * D + c) * 25 + k] = s;
        }
    if (c == 0) {
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float s = wave_sum(bs[o]);
            if (lane == 0) prow[nw + o] = s;
        }
    }
}

// token assembly backward, 16-byte loads issued up front (widths up to 1024: the vision towers'); same arithmetic as the
// kernel below, which keeps the wider rows
template <typename T, int MC>
__global__ __launch_bounds__(256) void embed_lnpre_bwd2_kernel(const T* __restrict__ dx, const T* __restrict__ patch,
                                                               const T* __restrict__ pos, const float* __restrict__ gamma,
                                                               T* __restrict__ dpatch, int B, int L, int width) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * L) return;
    const int b = row / L, l = row % L;
    if (l == 0) return;                                  // the class token is a frozen parameter
    const int nchunk = width >> 3;
    const T* src = patch + ((size_t)b * (L - 1) + (l - 1)) * width;
    const T* pr = pos + (size_t)l * width;
    const T* dr = dx + (size_t)row * width;
    float v[MC][8], gy[MC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float a[8], p[8], g[8];
            Vec8<T>::load(src + c * 8, a);
            Vec8<T>::load(pr + c * 8, p);
            Vec8<T>::load(dr + c * 8, gy[i]);
            load8f(gamma + c * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = Elem<T>::to_f(Elem<T>::from_f(a[e] + p[e]));
                s += v[i][e];
                gy[i][e] *= g[e];
            }
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MC; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (v[i][e] - mean) * rstd;
                s1 += gy[i][e];
                s2 += gy[i][e] * v[i][e];
            }
        }
    const float m1 = wave_sum(s1) / (float)width, m2 = wave_sum(s2) / (float)width;
    T* dst = dpatch + ((size_t)b * (L - 1) + (l - 1)) * width;
#pragma unroll
    for (int i = 0; i < MC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (gy[i][e] - m1 - v[i][e] * m2);
            Vec8<T>::store(dst + c * 8, o);
        }
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

// partial {min, max} pairs per ViT image the forward pass writes / partial weight-gradient rows per image of ffm_slice_bwd
extern "C" int ffm_slice_blocks(int H, int W) { return ((H + FW_R - 1) / FW_R) * ((W + SW - 1) / SW); }
extern "C" int ffm_slice_wgrad_blocks(int H, int W) { return ((H + WG_R - 1) / WG_R) * ((W + SW - 1) / SW); }

extern "C" int ffm_slice_conv_fwd(const float* img, const float* w, const float* bias, float* conv, float* mm_part,
                                  float* mnmx, int32_t* cnt, int N, int D, int H, int W, void* stream) {
    if (!img || !w || !bias || !conv || !mm_part || !mnmx || !cnt || N <= 0 || D <= 0 || D > MAXD || H <= 0 || W <= 0)
        return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nrb = (H + FW_R - 1) / FW_R, strips = (W + SW - 1) / SW;
    dim3 grid(strips, (nrb + 3) / 4, N);
    hipLaunchKernelGGL(slice_conv_fwd2_kernel, grid, dim3(256), 0, s, img, w, bias, conv, mm_part, D, H, W, nrb);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(minmax_reduce_kernel, dim3(N), dim3(256), 0, s, mm_part, mnmx, cnt, strips * nrb);
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
    if (dtype == FFM_BF16 && width % 8 == 0 && width <= 1024) {
        if (width <= 512)
            hipLaunchKernelGGL((embed_lnpre_bwd2_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)dx, (const bf16_t*)patch,
                               (const bf16_t*)pos, gamma, (bf16_t*)dpatch, B, L, width);
        else
            hipLaunchKernelGGL((embed_lnpre_bwd2_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)dx, (const bf16_t*)patch,
                               (const bf16_t*)pos, gamma, (bf16_t*)dpatch, B, L, width);
    } else if (dtype == FFM_BF16)
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
    const int nrb = (H + WG_R - 1) / WG_R, cg = (D + 3) / 4;
    dim3 g3((W + SW - 1) / SW, nrb, N * cg);
    hipLaunchKernelGGL(slice_conv_wgrad2_kernel, g3, dim3(256), 0, s, img, conv, dconv, mnmx, gmm, wpart, D, H, W, nrb, cg);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_slice_bwd_ab_blocks(void) { return SLICE_BWD_BLOCKS; }
