// RN50 trunk building blocks on NHWC rows (row = (b, y, x), C contiguous): everything of ModifiedResNet that is not a
// 1x1 convolution (those are ffm_gemm_nt on the rows, with the FairLoRA epilogue).  clip/model.py:11-118, 227-301.
//   3x3 convolutions (frozen weights): im2col -> ffm_gemm_nt with W as [Cout, (ky,kx,c)] -> (backward) dcols = dY W,
//                                      col2im by gathering the <= 9 taps of every input pixel (no atomics)
//   BatchNorm2d, TRAIN mode, trainable: two-stage column sums, finalize in double, normalise (+ReLU); backward likewise
//   AvgPool2d(2), residual add + ReLU, attention-pool token assembly (mean token + positional embedding)
// All of it is HBM-bound elementwise / reduction work; 16-byte accesses along C.
#include "common.h"
#include <cstdlib>

namespace {

template <typename T> struct VecC;             // 16-byte chunk of channels <-> floats
template <> struct VecC<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) { Vec8<bf16_t>::load(p, v); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) { Vec8<bf16_t>::store(p, v); }
};
template <> struct VecC<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    }
};

inline int grid1d(size_t n, int cap = 16384) {
    size_t b = (n + 255) / 256;
    return (int)(b > (size_t)cap ? cap : (b < 1 ? 1 : b));
}

// ---- stem conv1: raw NCHW fp32 image -> normalised 3x3 / pad 1 patches, k = (ky*3 + kx)*3 + c, zero padded to Kp
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int B, int H,
                                                          int W, int stride, int Kp, f32x4 mean3, f32x4 std3) {
    constexpr int VN = VecC<T>::N;                                   // one 16-byte store per thread
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1, kc = Kp / VN;
    const size_t total = (size_t)B * Ho * Wo * kc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(i % kc) * VN;
        size_t r = i / kc;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const int k = k0 + e;
            v[e] = 0.f;
            if (k < 27) {
                const int c = k % 3, tap = k / 3, ky = tap / 3, kx = tap % 3;
                const int y = oy * stride + ky - 1, x = ox * stride + kx - 1;
                if (y >= 0 && y < H && x >= 0 && x < W)
                    v[e] = (img[(((size_t)b * 3 + c) * H + y) * W + x] / 255.0f - mean3[c]) / std3[c];
            }
        }
        VecC<T>::store(cols + i * VN, v);
    }
}

// ---- generic 3x3 / pad 1: x [B*H*W, C] -> cols [B*Ho*Wo, Kp], k = (ky*3 + kx)*C + c
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ cols, int B, int H, int W,
                                                        int C, int stride, int Kp) {
    constexpr int VN = VecC<T>::N;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1, kc = Kp / VN;
    const size_t total = (size_t)B * Ho * Wo * kc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % kc) * VN;
        size_t r = i / kc;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = 0.f;
        if (k < 9 * C) {
            const int tap = k / C, c = k % C, ky = tap / 3, kx = tap % 3;
            const int y = oy * stride + ky - 1, xx = ox * stride + kx - 1;
            if (y >= 0 && y < H && xx >= 0 && xx < W) VecC<T>::load(x + (((size_t)b * H + y) * W + xx) * C + c, v);
        }
        VecC<T>::store(cols + ((((size_t)b * Ho + oy) * Wo + ox)) * Kp + k, v);
    }
}

// ---- backward of the above: dx[b,y,x,c] = sum over the taps whose output position exists
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* __restrict__ dcols, T* __restrict__ dx, int B, int H, int W,
                                                        int C, int stride, int Kp) {
    constexpr int VN = VecC<T>::N;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1, cc = C / VN;
    const size_t total = (size_t)B * H * W * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * VN;
        size_t r = i / cc;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = y + 1 - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = x + 1 - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                float v[VN];
                VecC<T>::load(dcols + (((size_t)b * Ho + oy) * Wo + ox) * Kp + (ky * 3 + kx) * C + c, v);
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] += v[e];
            }
        }
        VecC<T>::store(dx + i * VN, acc);
    }
}

// ---- column sums over the rows: part[blk][0][c] = sum a, part[blk][1][c] = sum a*b
//  MODE 0: a = x, b = x                                (BatchNorm statistics)
//  MODE 1: a = g, b = g * xhat, g = dy * (y > 0 if mask) (BatchNorm backward)
// A block owns rows [blk*rpb, (blk+1)*rpb) and up to 256 16-byte channel chunks: min(C/VN, 256) threads lie along a
// row (coalesced 16-byte loads), the remaining threads are row lanes; the row lanes are combined through LDS.
constexpr int CS_MAXBLK = 1024;  // upper bound of the partial rows (the finalize kernels sum them 8 lanes per channel)
// rows per block: 64 for the large maps, fewer for the deep layers' small ones (7 x 7 maps at batch 32 are 1568 rows:
// 64-row blocks were 25 blocks on 256 CUs, 24.8 us per launch) so that a launch has ~256 blocks or more
inline int cs_blocks(int rows) {
    int rpb = rows / 256;
    rpb = rpb < 8 ? 8 : (rpb > 64 ? 64 : rpb);
    const int n = (rows + rpb - 1) / rpb;
    return n < 1 ? 1 : (n > CS_MAXBLK ? CS_MAXBLK : n);
}
template <typename T, int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ p0, const T* __restrict__ p1,
                                                     const T* __restrict__ mask, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ part, int rows,
                                                     int C, int rpb, T* __restrict__ gout = nullptr, int strip = 0) {
    // strip > 0: a block owns `strip` channel chunks (a 128-byte strip of every row) instead of up to 256 - the geometry
    // of the folded BatchNorm path below, whose few partial rows want many blocks along the channels
    constexpr int VN = VecC<T>::N;
    __shared__ float red[2][256][VN + 1];
    const int cc = C / VN;
    const int tpr = strip > 0 ? (cc < strip ? cc : strip) : (cc < 256 ? cc : 256), nrl = 256 / tpr;
    const int chl = threadIdx.x % tpr, rl = threadIdx.x / tpr;
    const int ch = blockIdx.y * tpr + chl;
    const int r0 = blockIdx.x * rpb, r1 = (r0 + rpb) < rows ? (r0 + rpb) : rows;
    float s0[VN], s1[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) s0[e] = s1[e] = 0.f;
    if (rl < nrl && ch < cc) {
        float mu[VN], rs[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            mu[e] = MODE == 1 ? mean[ch * VN + e] : 0.f;
            rs[e] = MODE == 1 ? rstd[ch * VN + e] : 0.f;
        }
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += nrl) {
            const size_t o = (size_t)r * C + (size_t)ch * VN;
            float a[VN];
            VecC<T>::load(p0 + o, a);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < VN; ++e) { s0[e] += a[e]; s1[e] += a[e] * a[e]; }
            } else {
                float xv[VN], m[VN];
                VecC<T>::load(p1 + o, xv);
                if (mask) VecC<T>::load(mask + o, m);
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    const float g = (mask && !(m[e] > 0.f)) ? 0.f : a[e];
                    s0[e] += g;
                    s1[e] += g * (xv[e] - mu[e]) * rs[e];
                    a[e] = g;
                }
                if (gout) VecC<T>::store(gout + o, a);       // g = dy * (y > 0): the identity path's gradient (relu_bwd)
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) { red[0][threadIdx.x][e] = s0[e]; red[1][threadIdx.x][e] = s1[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < tpr * VN; i += 256) {
        const int cl = i / VN, e = i % VN;
        if (blockIdx.y * tpr + cl >= cc) continue;
        float t0 = 0.f, t1 = 0.f;
        for (int l = 0; l < nrl; ++l) { t0 += red[0][l * tpr + cl][e]; t1 += red[1][l * tpr + cl][e]; }
        const int c = (blockIdx.y * tpr + cl) * VN + e;
        part[((size_t)blockIdx.x * 2 + 0) * C + c] = t0;
        part[((size_t)blockIdx.x * 2 + 1) * C + c] = t1;
    }
}

// totals over nb partial rows [nb][2][C] of the strip's channels [c0, c0 + sw): tot[j * 64 + cl], j = 0 (sum a) / 1 (sum a b).
// sw <= 64, a multiple of 4; 4 adjacent channels per thread (16-byte loads), 256 / (sw / 2) threads share them along nb.
__device__ __forceinline__ void bnf_head_totals(const float* __restrict__ part, int nb, int C, int c0, int sw, double* tot,
                                                double (*red)[4]) {
    const int nq = sw / 2;                                      // quads of channels x 2 sums
    const int tq = 256 / nq;
    const int quad = threadIdx.x % nq, sub = threadIdx.x / nq;
    const int j = quad / (sw / 4), c = c0 + (quad % (sw / 4)) * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (sub < tq) {
#pragma unroll 8
        for (int b = sub; b < nb; b += tq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + ((size_t)b * 2 + j) * C + c);
            a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
        }
    }
    red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1; red[threadIdx.x][2] = a2; red[threadIdx.x][3] = a3;
    __syncthreads();
    if ((int)threadIdx.x < 2 * sw) {
        const int jj = threadIdx.x / sw, cl = threadIdx.x % sw;
        const int q = jj * (sw / 4) + cl / 4, e = cl % 4;
        double s = 0.0;
        for (int k = 0; k < tq; ++k) s += red[k * nq + q][e];
        tot[jj * 64 + cl] = s;
    }
    __syncthreads();
}

// The finalize launches of the three-launch path (large maps).  A block owns `sw` adjacent channels (16; 4 when there are
// more than 512 partial rows - layer1 and the stem, whose few channels would otherwise make a handful of blocks) and walks the partial rows with 16-byte loads, 256 / (sw / 2) threads side by
// side (round 4's kernel gave every channel 64 lanes with 4-byte loads: each touched a cache line of its own, 6.4 us).
__device__ __forceinline__ int fin_sw(int nblk) { return nblk > 512 ? 4 : 16; }
inline int fin_sw_host(int nblk) { return nblk > 512 ? 4 : 16; }

// mean / rstd of the batch (biased variance, eps 1e-5) + running statistics update (unbiased variance, momentum)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int rows, int C,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float momentum, float eps) {
    __shared__ double red[256][4];
    __shared__ double tot[128];
    const int sw0 = fin_sw(nblk), c0 = blockIdx.x * sw0, sw = (C - c0) < sw0 ? (C - c0) : sw0;
    bnf_head_totals(part, nblk, C, c0, sw, tot, red);
    if ((int)threadIdx.x >= sw) return;
    const int c = c0 + threadIdx.x;
    const double mu = tot[threadIdx.x] / rows;
    double var = tot[64 + threadIdx.x] / rows - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = rows > 1 ? var * rows / (rows - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
}

// eval mode: mean / rstd from the running statistics
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    rstd[c] = 1.0f / sqrtf(run_var[c] + eps);
}

// The apply kernels: threads along a row own FIXED channels (min(C/VN, 256) 16-byte chunks; blockIdx.y the chunk
// group), the remaining threads of the block and blockIdx.x walk the rows.  The per-channel parameters are read once per
// thread (as the flat one-chunk-per-thread loop had it, every 16 bytes of payload cost 32 scalar parameter loads and
// an integer modulo).
struct bn_lanes {
    int cc, tpr, nrl, ch, rl;
    bool on;
    template <int VN>
    __device__ __forceinline__ static bn_lanes make(int C) {
        bn_lanes l;
        l.cc = C / VN;
        l.tpr = l.cc < 256 ? l.cc : 256;
        l.nrl = 256 / l.tpr;
        l.ch = blockIdx.y * 256 + threadIdx.x % l.tpr;
        l.rl = threadIdx.x / l.tpr;
        l.on = l.rl < l.nrl && l.ch < l.cc;
        return l;
    }
};
template <int VN>
__device__ __forceinline__ void bn_param(const float* __restrict__ p, int c0, float (&v)[VN]) {
#pragma unroll
    for (int q = 0; q < VN / 4; ++q) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + c0 + 4 * q);
        v[4 * q] = a[0]; v[4 * q + 1] = a[1]; v[4 * q + 2] = a[2]; v[4 * q + 3] = a[3];
    }
}

// y = (x - mean) * rstd * gamma + beta  (+ residual) (ReLU)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res,
                                                       T* __restrict__ y, int rows, int C, int relu) {
    constexpr int VN = VecC<T>::N;
    const bn_lanes L = bn_lanes::make<VN>(C);
    if (!L.on) return;
    float mu[VN], rs[VN], ga[VN], be[VN];
    bn_param<VN>(mean, L.ch * VN, mu);
    bn_param<VN>(rstd, L.ch * VN, rs);
    bn_param<VN>(gamma, L.ch * VN, ga);
    bn_param<VN>(beta, L.ch * VN, be);
    for (int r = blockIdx.x * L.nrl + L.rl; r < rows; r += gridDim.x * L.nrl) {
        const size_t o = (size_t)r * C + (size_t)L.ch * VN;
        float v[VN], rr[VN];
        VecC<T>::load(x + o, v);
        if (res) VecC<T>::load(res + o, rr);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float t = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
            if (res) t += rr[e];
            v[e] = relu ? fmaxf(t, 0.f) : t;
        }
        VecC<T>::store(y + o, v);
    }
}

// dgamma = sum g xhat, dbeta = sum g; k1 = sum g / N, k2 = sum g xhat / N for the apply pass
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int rows, int C,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ k12) {
    __shared__ double red[256][4];
    __shared__ double tot[128];
    const int sw0 = fin_sw(nblk), c0 = blockIdx.x * sw0, sw = (C - c0) < sw0 ? (C - c0) : sw0;
    bnf_head_totals(part, nblk, C, c0, sw, tot, red);
    if ((int)threadIdx.x >= sw) return;
    const int c = c0 + threadIdx.x;
    const double s = tot[threadIdx.x], q = tot[64 + threadIdx.x];
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    k12[c] = (float)(s / rows);
    k12[C + c] = (float)(q / rows);
}

// dx = gamma * rstd * (g - k1 - xhat * k2), g = dy * (mask > 0)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ mask,
                                                           const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ k12, T* __restrict__ dx, int rows,
                                                           int C) {
    constexpr int VN = VecC<T>::N;
    const bn_lanes L = bn_lanes::make<VN>(C);
    if (!L.on) return;
    float mu[VN], rs[VN], ga[VN], k1[VN], k2[VN];
    bn_param<VN>(mean, L.ch * VN, mu);
    bn_param<VN>(rstd, L.ch * VN, rs);
    bn_param<VN>(gamma, L.ch * VN, ga);
    bn_param<VN>(k12, L.ch * VN, k1);
    bn_param<VN>(k12 + C, L.ch * VN, k2);
    for (int r = blockIdx.x * L.nrl + L.rl; r < rows; r += gridDim.x * L.nrl) {
        const size_t o = (size_t)r * C + (size_t)L.ch * VN;
        float g[VN], xv[VN], m[VN];
        VecC<T>::load(dy + o, g);
        VecC<T>::load(x + o, xv);
        if (mask) VecC<T>::load(mask + o, m);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float gg = g[e];
            if (mask && !(m[e] > 0.f)) gg = 0.f;
            const float xh = (xv[e] - mu[e]) * rs[e];
            g[e] = ga[e] * rs[e] * (gg - k1[e] - xh * k2[e]);
        }
        VecC<T>::store(dx + o, g);
    }
}


// ---- BatchNorm on the small maps without the finalize launch ("folded": layer3 / layer4, rows <= FFM_BN_FOLD_ROWS) ----
// The three-launch path spends a launch of its own (6.4 us + a kernel boundary) on summing the partial rows, 110 times per
// RN50 step.  Here every block of the apply pass sums the partial rows of ITS channels at its head, in a fixed order (all
// blocks get the same bits), and the blocks of row group 0 write the per-channel results.  That only pays while the
// partials are few and the blocks own few channels - head traffic = row groups x partial rows x 8 C bytes - so the
// geometry is a strip one: a block owns 128 bytes of every row (64 / 32 channels) and one of G row groups; the backward
// column-sum pass runs on the same strips with G partial rows (bn_fold_geom).  The large maps keep the three launches.
// Measured on the RN50 step (bs 32, one call): 6.53 ms without, 6.29 with rows <= 8192 folded, 6.26 with layer2 as well.
constexpr int BNF_STRIP_BYTES = 128;
struct bn_fold { bool on; int G, strip; };
inline bn_fold bn_fold_geom(int rows, int C, int vn, int given_part_rows) {
    static const int max_rows = getenv("FFM_BN_FOLD_ROWS") ? atoi(getenv("FFM_BN_FOLD_ROWS")) : 32768;
    bn_fold f{false, 0, BNF_STRIP_BYTES / 16};
    if (rows > max_rows) return f;
    int G = 1;
    while ((long long)(G + 1) * (G + 1) * 8 * C <= (16ll << 20) && G < 128) ++G;     // G^2 x 8 C bytes of head traffic <= 16 MB
    const int most = cs_blocks(rows);                                                  // (the partial buffer is sized by it)
    if (G > most) G = most;
    if (G > rows) G = rows;
    if (given_part_rows > 0) {                                                         // forward: the producer's row tiles
        if (given_part_rows > 1024) return f;
        long long g2 = (16ll << 20) / ((long long)given_part_rows * 8 * C);
        if (g2 < 8) return f;
        if (g2 < G) G = (int)g2;
    }
    // a mid-sized map (layer2: 25 088 rows) with few channels makes too few strips to stream at the HBM rate (C = 128:
    // 256 blocks, the backward apply pass 11.4 us against 6.4 + a 6.7 us finalize launch): those keep the three launches
    if (rows > 8192 && (long long)G * ((C / vn + f.strip - 1) / f.strip) < 384) return f;
    f.on = true;
    f.G = G < 1 ? 1 : G;
    return f;
}

struct bnf_lanes {
    int tpr, nrl, ch, rl, c0, sw;
    bool on;
    template <int VN>
    __device__ __forceinline__ static bnf_lanes make(int C, int strip) {
        bnf_lanes l;
        const int cc = C / VN;
        l.tpr = cc < strip ? cc : strip;
        l.nrl = 256 / l.tpr;
        l.ch = blockIdx.y * l.tpr + threadIdx.x % l.tpr;
        l.rl = threadIdx.x / l.tpr;
        l.on = l.rl < l.nrl && l.ch < cc;
        l.c0 = blockIdx.y * l.tpr * VN;
        l.sw = (C - l.c0) < l.tpr * VN ? (C - l.c0) : l.tpr * VN;
        return l;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fold_kernel(const T* __restrict__ x, const float* __restrict__ part, int nb,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const T* __restrict__ res, T* __restrict__ y, int rows, int C,
                                                            int relu, int strip, float momentum, float eps) {
    constexpr int VN = VecC<T>::N;
    __shared__ double red[256][4];
    __shared__ double tot[128];
    __shared__ float par[2][64];                                // mean, rstd of the strip
    const bnf_lanes L = bnf_lanes::make<VN>(C, strip);
    bnf_head_totals(part, nb, C, L.c0, L.sw, tot, red);
    if ((int)threadIdx.x < L.sw) {
        const int c = L.c0 + threadIdx.x;
        const double mu = tot[threadIdx.x] / rows;
        double var = tot[64 + threadIdx.x] / rows - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m = (float)mu, r = (float)(1.0 / sqrt(var + (double)eps));
        par[0][threadIdx.x] = m;
        par[1][threadIdx.x] = r;
        if (blockIdx.x == 0) {                                  // one writer per channel
            mean[c] = m;
            rstd[c] = r;
            if (run_mean) {
                const double unb = rows > 1 ? var * rows / (rows - 1) : var;
                run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
                run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
            }
        }
    }
    __syncthreads();
    if (!L.on) return;
    float mu[VN], rs[VN], ga[VN], be[VN];
    const int cl = (L.ch * VN) - L.c0;
#pragma unroll
    for (int e = 0; e < VN; ++e) { mu[e] = par[0][cl + e]; rs[e] = par[1][cl + e]; }
    bn_param<VN>(gamma, L.ch * VN, ga);
    bn_param<VN>(beta, L.ch * VN, be);
#pragma unroll 4
    for (int r = blockIdx.x * L.nrl + L.rl; r < rows; r += gridDim.x * L.nrl) {
        const size_t o = (size_t)r * C + (size_t)L.ch * VN;
        float v[VN], rr[VN];
        VecC<T>::load(x + o, v);
        if (res) VecC<T>::load(res + o, rr);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float t = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
            if (res) t += rr[e];
            v[e] = relu ? fmaxf(t, 0.f) : t;
        }
        VecC<T>::store(y + o, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_fold_kernel(const T* __restrict__ dy, const T* __restrict__ mask,
                                                                const T* __restrict__ x, const float* __restrict__ part, int nb,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, T* __restrict__ dx, int rows, int C,
                                                                int strip) {
    constexpr int VN = VecC<T>::N;
    __shared__ double red[256][4];
    __shared__ double tot[128];
    __shared__ float par[2][64];                                // k1 = sum g / N, k2 = sum g xhat / N
    const bnf_lanes L = bnf_lanes::make<VN>(C, strip);
    bnf_head_totals(part, nb, C, L.c0, L.sw, tot, red);
    if ((int)threadIdx.x < L.sw) {
        const int c = L.c0 + threadIdx.x;
        const double s = tot[threadIdx.x], q = tot[64 + threadIdx.x];
        par[0][threadIdx.x] = (float)(s / rows);
        par[1][threadIdx.x] = (float)(q / rows);
        if (blockIdx.x == 0) {
            dbeta[c] = (float)s;
            dgamma[c] = (float)q;
        }
    }
    __syncthreads();
    if (!L.on) return;
    float mu[VN], rs[VN], ga[VN], k1[VN], k2[VN];
    const int cl = (L.ch * VN) - L.c0;
#pragma unroll
    for (int e = 0; e < VN; ++e) { k1[e] = par[0][cl + e]; k2[e] = par[1][cl + e]; }
    bn_param<VN>(mean, L.ch * VN, mu);
    bn_param<VN>(rstd, L.ch * VN, rs);
    bn_param<VN>(gamma, L.ch * VN, ga);
#pragma unroll 2
    for (int r = blockIdx.x * L.nrl + L.rl; r < rows; r += gridDim.x * L.nrl) {
        const size_t o = (size_t)r * C + (size_t)L.ch * VN;
        float g[VN], xv[VN], m[VN];
        VecC<T>::load(dy + o, g);
        VecC<T>::load(x + o, xv);
        if (mask) VecC<T>::load(mask + o, m);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float gg = g[e];
            if (mask && !(m[e] > 0.f)) gg = 0.f;
            const float xh = (xv[e] - mu[e]) * rs[e];
            g[e] = ga[e] * rs[e] * (gg - k1[e] - xh * k2[e]);
        }
        VecC<T>::store(dx + o, g);
    }
}

// ---- AvgPool2d(2) on NHWC
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W,
                                                       int C) {
    constexpr int VN = VecC<T>::N;
    const int Ho = H / 2, Wo = W / 2, cc = C / VN;
    // forward: one thread per pooled element chunk; backward: one thread per input element chunk
    const size_t total = BWD ? (size_t)B * H * W * cc : (size_t)B * Ho * Wo * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * VN;
        size_t r = i / cc;
        float v[VN];
        if (!BWD) {
            const int ox = (int)(r % Wo); r /= Wo;
            const int oy = (int)(r % Ho);
            const int b = (int)(r / Ho);
            float acc[VN];
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    VecC<T>::load(in + (((size_t)b * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c, v);
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[e] += v[e];
                }
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] *= 0.25f;
            VecC<T>::store(out + i * VN, acc);
        } else {
            const int x = (int)(r % W); r /= W;
            const int y = (int)(r % H);
            const int b = (int)(r / H);
            VecC<T>::load(in + (((size_t)b * Ho + y / 2) * Wo + x / 2) * C + c, v);     // in = d(pooled)
#pragma unroll
            for (int e = 0; e < VN; ++e) v[e] *= 0.25f;
            VecC<T>::store(out + i * VN, v);
        }
    }
}

// out = a + b  (gradient accumulation of the two branches of a Bottleneck)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                  size_t n) {
    constexpr int VN = VecC<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VN; i += (size_t)gridDim.x * blockDim.x) {
        float x[VN], y[VN];
        VecC<T>::load(a + i * VN, x);
        VecC<T>::load(b + i * VN, y);
#pragma unroll
        for (int e = 0; e < VN; ++e) x[e] += y[e];
        VecC<T>::store(out + i * VN, x);
    }
}

// out = g * (y > 0): gradient through a ReLU whose output is y
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ out,
                                                       size_t n) {
    constexpr int VN = VecC<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VN; i += (size_t)gridDim.x * blockDim.x) {
        float a[VN], m[VN];
        VecC<T>::load(g + i * VN, a);
        VecC<T>::load(y + i * VN, m);
#pragma unroll
        for (int e = 0; e < VN; ++e) a[e] = m[e] > 0.f ? a[e] : 0.f;
        VecC<T>::store(out + i * VN, a);
    }
}

// ---- attention pool tokens: tok[b,0] = mean_hw x[b,hw] + pos[0]; tok[b,1+hw] = x[b,hw] + pos[1+hw]  (clip/model.py:76-78)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const T* __restrict__ in, const T* __restrict__ pos,
                                                              T* __restrict__ out, int B, int HW, int E) {
    constexpr int VN = VecC<T>::N;
    const int ec = E / VN, L = HW + 1;
    const size_t total = BWD ? (size_t)B * HW * ec : (size_t)B * L * ec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e0 = (int)(i % ec) * VN;
        size_t r = i / ec;
        float v[VN], acc[VN];
        if (!BWD) {
            const int l = (int)(r % L), b = (int)(r / L);
            if (l == 0) {
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = 0.f;
                for (int hw = 0; hw < HW; ++hw) {
                    VecC<T>::load(in + ((size_t)b * HW + hw) * E + e0, v);
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[e] += v[e];
                }
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] /= (float)HW;
            } else {
                VecC<T>::load(in + ((size_t)b * HW + l - 1) * E + e0, acc);
            }
            // the mean token is rounded to the activation dtype before the positional embedding is added, as
            // torch.cat([x.mean(0), x]) + pos does
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] = Elem<T>::to_f(Elem<T>::from_f(acc[e]));
            VecC<T>::load(pos + (size_t)l * E + e0, v);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] += v[e];
            VecC<T>::store(out + i * VN, acc);
        } else {
            const int hw = (int)(r % HW), b = (int)(r / HW);        // in = d(tokens) [B*L, E]
            VecC<T>::load(in + ((size_t)b * L + 1 + hw) * E + e0, acc);
            VecC<T>::load(in + ((size_t)b * L) * E + e0, v);
#pragma unroll
            for (int e = 0; e < VN; ++e) acc[e] += v[e] / (float)HW;
            VecC<T>::store(out + i * VN, acc);
        }
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
    if ((dtype) == FFM_BF16) { CALL_BF16; } else if ((dtype) == FFM_F32) { CALL_F32; } else return FFM_EINVAL;

extern "C" int ffm_stem_im2col(const float* img, void* cols, int B, int H, int W, int stride, int Kp, const float* mean3,
                               const float* std3, int dtype, void* stream) {
    if (!img || !cols || !mean3 || !std3 || B <= 0 || H <= 0 || W <= 0 || stride <= 0 || Kp < 27) return FFM_EINVAL;
    if (Kp % (dtype == FFM_BF16 ? 8 : 4) || ((uintptr_t)cols & 15)) return FFM_EINVAL;
    const f32x4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int g = grid1d((size_t)B * Ho * Wo * Kp / (dtype == FFM_BF16 ? 8 : 4));
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((stem_im2col_kernel<bf16_t>), dim3(g), dim3(256), 0, s, img, (bf16_t*)cols, B, H, W, stride, Kp, m, sd),
               hipLaunchKernelGGL((stem_im2col_kernel<float>), dim3(g), dim3(256), 0, s, img, (float*)cols, B, H, W, stride, Kp, m, sd))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

static bool conv_args_ok(const void* a, const void* b, int B, int H, int W, int C, int stride, int Kp, int dtype) {
    const int vn = dtype == FFM_BF16 ? 8 : 4;
    return a && b && B > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && C % vn == 0 && Kp >= 9 * C &&
           Kp % vn == 0 && !(((uintptr_t)a | (uintptr_t)b) & 15);
}

extern "C" int ffm_im2col3x3(const void* x, void* cols, int B, int H, int W, int C, int stride, int Kp, int dtype,
                             void* stream) {
    if (!conv_args_ok(x, cols, B, H, W, C, stride, Kp, dtype)) return FFM_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int g = grid1d((size_t)B * Ho * Wo * Kp / 4);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((im2col3x3_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)cols, B, H, W, C, stride, Kp),
               hipLaunchKernelGGL((im2col3x3_kernel<float>), dim3(g), dim3(256), 0, s, (const float*)x, (float*)cols, B, H, W, C, stride, Kp))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_col2im3x3(const void* dcols, void* dx, int B, int H, int W, int C, int stride, int Kp, int dtype,
                             void* stream) {
    if (!conv_args_ok(dcols, dx, B, H, W, C, stride, Kp, dtype)) return FFM_EINVAL;
    const int g = grid1d((size_t)B * H * W * C / 4);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((col2im3x3_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)dcols, (bf16_t*)dx, B, H, W, C, stride, Kp),
               hipLaunchKernelGGL((col2im3x3_kernel<float>), dim3(g), dim3(256), 0, s, (const float*)dcols, (float*)dx, B, H, W, C, stride, Kp))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

namespace {
// grid of the apply kernels: blockIdx.y = group of 256 channel chunks, blockIdx.x walks the rows; FFM_BN_RPT rows per
// thread (default 2; 1 / 2 / 4 / 8 measured 7.09 / 7.04 / 7.05 / 7.24 ms on the RN50 step)
inline dim3 bn_apply_grid(int rows, int C, int vn) {
    static const int rpt = getenv("FFM_BN_RPT") ? atoi(getenv("FFM_BN_RPT")) : 2;
    const int cc = C / vn, tpr = cc < 256 ? cc : 256, nrl = 256 / tpr, gy = (cc + 255) / 256;
    int gx = (rows + nrl * rpt - 1) / (nrl * rpt);
    const int cap = 65535;
    return dim3(gx < 1 ? 1 : (gx > cap ? cap : gx), gy);
}
}  // namespace

extern "C" int ffm_bn_blocks(int rows) { return cs_blocks(rows); }

extern "C" int ffm_bn_fwd(const void* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                          float* mean, float* rstd, float* part, int part_rows, const void* res, void* y, int rows, int C,
                          int training, int relu, int dtype, void* stream) {
    if (!x || !gamma || !beta || !run_mean || !run_var || !mean || !rstd || !y || rows <= 0 || C <= 0) return FFM_EINVAL;
    if (C % (dtype == FFM_BF16 ? 8 : 4) || (training && !part) || ((uintptr_t)part & 15)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int vn = dtype == FFM_BF16 ? 8 : 4;
    const bn_fold fold = training ? bn_fold_geom(rows, C, vn, part_rows) : bn_fold{false, 0, 0};
    if (fold.on) {                                   // small map: no finalize launch (bn_apply_fold_kernel sums the partials)
        if (((uintptr_t)part & 15) || C % 4) return FFM_EINVAL;
        const int gy = (C / vn + fold.strip - 1) / fold.strip;
        int nb = part_rows;
        if (part_rows <= 0) {
            const int rpb = (rows + fold.G - 1) / fold.G;
            nb = (rows + rpb - 1) / rpb;
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL((colsum_kernel<bf16_t, 0>), dim3(nb, gy), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, part, rows, C, rpb, (bf16_t*)nullptr, fold.strip),
                       hipLaunchKernelGGL((colsum_kernel<float, 0>), dim3(nb, gy), dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, rows, C, rpb, (float*)nullptr, fold.strip))
            FFM_CHECK_LAUNCH();
        }
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((bn_apply_fold_kernel<bf16_t>), dim3(fold.G, gy), dim3(256), 0, s, (const bf16_t*)x, part, nb, mean, rstd, run_mean, run_var, gamma, beta, (const bf16_t*)res, (bf16_t*)y, rows, C, relu, fold.strip, 0.1f, 1e-5f),
                   hipLaunchKernelGGL((bn_apply_fold_kernel<float>), dim3(fold.G, gy), dim3(256), 0, s, (const float*)x, part, nb, mean, rstd, run_mean, run_var, gamma, beta, (const float*)res, (float*)y, rows, C, relu, fold.strip, 0.1f, 1e-5f))
        FFM_CHECK_LAUNCH();
        return FFM_OK;
    }
    if (training && part_rows > 0) {
        // the producer of x left the column sums of its row tiles behind (ffm_gemm_args.colstat_part)
        if (part_rows > 4096) return FFM_EINVAL;
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + fin_sw_host(part_rows) - 1) / fin_sw_host(part_rows)), dim3(256), 0, s, part, part_rows, rows, C, mean, rstd, run_mean,
                           run_var, 0.1f, 1e-5f);
    } else if (training) {
        const int nb0 = cs_blocks(rows), rpb = (rows + nb0 - 1) / nb0, nblk = (rows + rpb - 1) / rpb;
        dim3 g(nblk, (C / (dtype == FFM_BF16 ? 8 : 4) + 255) / 256);
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((colsum_kernel<bf16_t, 0>), g, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, part, rows, C, rpb),
                   hipLaunchKernelGGL((colsum_kernel<float, 0>), g, dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, rows, C, rpb))
        FFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + fin_sw_host(nblk) - 1) / fin_sw_host(nblk)), dim3(256), 0, s, part, nblk, rows, C, mean, rstd, run_mean,
                           run_var, 0.1f, 1e-5f);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, run_mean, run_var, mean, rstd, C, 1e-5f);
    }
    FFM_CHECK_LAUNCH();
    const dim3 g2 = bn_apply_grid(rows, C, dtype == FFM_BF16 ? 8 : 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((bn_apply_kernel<bf16_t>), g2, dim3(256), 0, s, (const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)res, (bf16_t*)y, rows, C, relu),
               hipLaunchKernelGGL((bn_apply_kernel<float>), g2, dim3(256), 0, s, (const float*)x, mean, rstd, gamma, beta, (const float*)res, (float*)y, rows, C, relu))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_bn_bwd(const void* dy, const void* relu_out, const void* x, const float* gamma, const float* mean,
                          const float* rstd, float* part, int part_rows, float* k12, float* dgamma, float* dbeta, void* dx,
                          void* g_out, int rows, int C, int dtype, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !part || !k12 || !dgamma || !dbeta || !dx || rows <= 0 || C <= 0)
        return FFM_EINVAL;
    if (C % (dtype == FFM_BF16 ? 8 : 4) || ((uintptr_t)part & 15)) return FFM_EINVAL;
    if (part_rows < 0 || part_rows > 4096 || (part_rows > 0 && g_out)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bn_fold fold = bn_fold_geom(rows, C, dtype == FFM_BF16 ? 8 : 4, part_rows);
    if (fold.on) {                                   // small map: column sums on strips, no finalize launch
        if (((uintptr_t)part & 15) || C % 4) return FFM_EINVAL;
        const int gy = (C / (dtype == FFM_BF16 ? 8 : 4) + fold.strip - 1) / fold.strip;
        const int rpb = (rows + fold.G - 1) / fold.G;
        int nb = part_rows;                              // (> 0: the producer of dy left the column sums of its row tiles)
        if (part_rows <= 0) {
            nb = (rows + rpb - 1) / rpb;
            DISPATCH_T(dtype,
                       hipLaunchKernelGGL((colsum_kernel<bf16_t, 1>), dim3(nb, gy), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)relu_out, mean, rstd, part, rows, C, rpb, (bf16_t*)g_out, fold.strip),
                       hipLaunchKernelGGL((colsum_kernel<float, 1>), dim3(nb, gy), dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)relu_out, mean, rstd, part, rows, C, rpb, (float*)g_out, fold.strip))
            FFM_CHECK_LAUNCH();
        }
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((bn_bwd_apply_fold_kernel<bf16_t>), dim3(fold.G, gy), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)relu_out, (const bf16_t*)x, part, nb, mean, rstd, gamma, dgamma, dbeta, (bf16_t*)dx, rows, C, fold.strip),
                   hipLaunchKernelGGL((bn_bwd_apply_fold_kernel<float>), dim3(fold.G, gy), dim3(256), 0, s, (const float*)dy, (const float*)relu_out, (const float*)x, part, nb, mean, rstd, gamma, dgamma, dbeta, (float*)dx, rows, C, fold.strip))
        FFM_CHECK_LAUNCH();
        return FFM_OK;
    }
    const int nb0 = cs_blocks(rows), rpb = (rows + nb0 - 1) / nb0;
    int nblk = part_rows;
    if (part_rows <= 0) {
        nblk = (rows + rpb - 1) / rpb;
        dim3 g(nblk, (C / (dtype == FFM_BF16 ? 8 : 4) + 255) / 256);
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((colsum_kernel<bf16_t, 1>), g, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)relu_out, mean, rstd, part, rows, C, rpb, (bf16_t*)g_out),
                   hipLaunchKernelGGL((colsum_kernel<float, 1>), g, dim3(256), 0, s, (const float*)dy, (const float*)x, (const float*)relu_out, mean, rstd, part, rows, C, rpb, (float*)g_out))
        FFM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + fin_sw_host(nblk) - 1) / fin_sw_host(nblk)), dim3(256), 0, s, part, nblk, rows, C, dgamma, dbeta, k12);
    FFM_CHECK_LAUNCH();
    const dim3 g2 = bn_apply_grid(rows, C, dtype == FFM_BF16 ? 8 : 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), g2, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)relu_out, (const bf16_t*)x, mean, rstd, gamma, k12, (bf16_t*)dx, rows, C),
               hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), g2, dim3(256), 0, s, (const float*)dy, (const float*)relu_out, (const float*)x, mean, rstd, gamma, k12, (float*)dx, rows, C))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_avgpool2(const void* in, void* out, int B, int H, int W, int C, int backward, int dtype, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C % (dtype == FFM_BF16 ? 8 : 4)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid1d((size_t)B * H * W * C / (backward ? 4 : 16));
    if (backward) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((avgpool2_kernel<bf16_t, true>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, H, W, C),
                   hipLaunchKernelGGL((avgpool2_kernel<float, true>), dim3(g), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C))
    } else {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((avgpool2_kernel<bf16_t, false>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, B, H, W, C),
                   hipLaunchKernelGGL((avgpool2_kernel<float, false>), dim3(g), dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C))
    }
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
    if (!a || !b || !out || n <= 0 || n % (dtype == FFM_BF16 ? 8 : 4)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid1d((size_t)n / 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((add_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (size_t)n),
               hipLaunchKernelGGL((add_kernel<float>), dim3(g), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, (size_t)n))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_relu_bwd(const void* g, const void* y, void* out, int64_t n, int dtype, void* stream) {
    if (!g || !y || !out || n <= 0 || n % (dtype == FFM_BF16 ? 8 : 4)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int gr = grid1d((size_t)n / 4);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL((relu_bwd_kernel<bf16_t>), dim3(gr), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)y, (bf16_t*)out, (size_t)n),
               hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(gr), dim3(256), 0, s, (const float*)g, (const float*)y, (float*)out, (size_t)n))
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

extern "C" int ffm_attnpool_tokens(const void* in, const void* pos, void* out, int B, int HW, int E, int backward, int dtype,
                                   void* stream) {
    if (!in || !out || (!backward && !pos) || B <= 0 || HW <= 0 || E % (dtype == FFM_BF16 ? 8 : 4)) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid1d((size_t)B * (HW + 1) * E / 4);
    if (backward) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((attnpool_tokens_kernel<bf16_t, true>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)pos, (bf16_t*)out, B, HW, E),
                   hipLaunchKernelGGL((attnpool_tokens_kernel<float, true>), dim3(g), dim3(256), 0, s, (const float*)in, (const float*)pos, (float*)out, B, HW, E))
    } else {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL((attnpool_tokens_kernel<bf16_t, false>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)pos, (bf16_t*)out, B, HW, E),
                   hipLaunchKernelGGL((attnpool_tokens_kernel<float, false>), dim3(g), dim3(256), 0, s, (const float*)in, (const float*)pos, (float*)out, B, HW, E))
    }
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
