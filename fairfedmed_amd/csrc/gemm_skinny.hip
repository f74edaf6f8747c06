// Skinny GEMM: C = epilogue(A * B^T) for M <= 64 rows (the text tower: 4 prompts x 10 tokens = 40 rows against
// 512..2048-wide frozen weights).  The 128x128 kernel gives such a product N/128 = 4..16 blocks, each streaming its
// 128-column slice of the weight serially: 10-22 us of pure latency per launch, 96 launches per step on the side
// stream.  Here a block owns 16 output columns and ALL rows; its 4 or 8 waves split K, every wave issues the loads of
// up to 8 K32 steps before the first MFMA, and the partial accumulators meet in LDS: N/16 = 32..128 blocks.
// Operand types: see SkOps; epilogues: none, bias, bias+residual, bias+GELU, dGELU.
#include "gemm_panel.h"
#include <cstdlib>

#ifndef FFM_SKINNY_CAP_DEFAULT
#define FFM_SKINNY_CAP_DEFAULT 0
#endif
// Diagnostic twins only (tools/side_abl.sh): 1 = no weight loads, 2 = no activation loads (results are garbage) - which of
// the text tower's two operand streams is it that slows the vision chain's kernels down?
#ifndef FFM_SKINNY_ABL
#define FFM_SKINNY_ABL 0
#endif
#ifndef FFM_SKINNY_NT_DEFAULT
#define FFM_SKINNY_NT_DEFAULT 2
#endif
#ifndef FFM_SKINNY_MINB_DEFAULT
#define FFM_SKINNY_MINB_DEFAULT 1
#endif

namespace {

constexpr int SK_COLS = 16, SK_MF = 4;

// Operand traits.  TA: activation / output element, TB: weight element in memory, X3: see below.
//   <bf16, bf16>        one bf16 MFMA per K32 step
//   <float, float>      exact f32 MFMAs (the fp32 parity mode), K16 per step
//   <float, float, X3>  f32 operands in memory, products on the bf16 matrix cores: both fragments are split into bf16
//                       hi + lo pairs and hi*hi + lo*hi + hi*lo is accumulated in f32 (3 MFMAs at the bf16 rate, 16
//                       significant bits per operand; the f32 MFMA runs at 1/16 of that rate).  This is the text tower
//                       beside a bf16 vision tower (FFM_F32_X3; engine.py says why it is not simply bf16).
template <typename TA, typename TB, bool X3> struct SkOps;
template <> struct SkOps<bf16_t, bf16_t, false> {
    static constexpr int KS = 32, UN = 8;
    struct afrag { bf16x8 v; };
    typedef bf16x8 braw;
    typedef bf16x8 bprep;
    static __device__ __forceinline__ void loadA(afrag& f, const bf16_t* p) { f.v = *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ void loadB(braw& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ bprep prep(const braw& b) { return b; }
    static __device__ __forceinline__ void mma(f32x4& acc, const afrag& a, const bprep& b) { Mma16<bf16_t>::mma(acc, a.v, b); }
};
template <> struct SkOps<float, float, false> {
    static constexpr int KS = 16, UN = 8;
    struct afrag { f32x4 v; };
    typedef f32x4 braw;
    typedef f32x4 bprep;
    static __device__ __forceinline__ void loadA(afrag& f, const float* p) { f.v = *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void loadB(braw& f, const float* p) { f = *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ bprep prep(const braw& b) { return b; }
    static __device__ __forceinline__ void mma(f32x4& acc, const afrag& a, const bprep& b) { Mma16<float>::mma(acc, a.v, b); }
};
template <> struct SkOps<float, float, true> {
    static constexpr int KS = 32, UN = 4;
    struct afrag { f32x4 lo, hi; };                         // k 0..3 and 4..7 of the lane's group
    typedef afrag braw;
    struct bprep { bf16x8 h, l; };
    static __device__ __forceinline__ void loadA(afrag& f, const float* p) {
        f.lo = *reinterpret_cast<const f32x4*>(p);
        f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    }
    static __device__ __forceinline__ void loadB(braw& f, const float* p) { loadA(f, p); }
    static __device__ __forceinline__ bprep prep(const braw& a) {
        bprep r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r.h[i] = (bf16_t)a.lo[i];
            r.h[4 + i] = (bf16_t)a.hi[i];
            r.l[i] = (bf16_t)(a.lo[i] - (float)r.h[i]);
            r.l[4 + i] = (bf16_t)(a.hi[i] - (float)r.h[4 + i]);
        }
        return r;
    }
    static __device__ __forceinline__ void mma_split(f32x4& acc, const bprep& x, const bprep& b) {
        Mma16<bf16_t>::mma(acc, x.l, b.h);                  // small terms first
        Mma16<bf16_t>::mma(acc, x.h, b.l);
        Mma16<bf16_t>::mma(acc, x.h, b.h);
    }
    static __device__ __forceinline__ void mma(f32x4& acc, const afrag& a, const bprep& b) { mma_split(acc, prep(a), b); }
};

// <float, half, X3>: FFM_F32_X3_W16 - the weight rounded to IEEE half in memory (half the bytes), widened and split into the
// same bf16 hi + lo pair in the kernel (exact: 11 significant bits = 8 + 3), the activations as under X3
typedef _Float16 sk_half8 __attribute__((ext_vector_type(8)));
template <> struct SkOps<float, _Float16, true> {
    typedef SkOps<float, float, true> F;
    static constexpr int KS = F::KS, UN = F::UN;
    typedef F::afrag afrag;
    typedef sk_half8 braw;
    typedef F::bprep bprep;
    static __device__ __forceinline__ void loadA(afrag& f, const float* p) { F::loadA(f, p); }
    static __device__ __forceinline__ void loadB(braw& f, const _Float16* p) { f = *reinterpret_cast<const sk_half8*>(p); }
    static __device__ __forceinline__ bprep prep(const afrag& a) { return F::prep(a); }
    static __device__ __forceinline__ bprep prep(const braw& b) {
        bprep r;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = (float)b[i];
            r.h[i] = (bf16_t)x;
            r.l[i] = (bf16_t)(x - (float)r.h[i]);
        }
        return r;
    }
    static __device__ __forceinline__ void mma_split(f32x4& acc, const bprep& x, const bprep& b) { F::mma_split(acc, x, b); }
    static __device__ __forceinline__ void mma(f32x4& acc, const afrag& a, const bprep& b) { F::mma_split(acc, F::prep(a), b); }
};

template <typename TA, typename TB, bool X3, int NW, int FL>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(ffm_gemm_args p) {
    typedef SkOps<TA, TB, X3> O;
    constexpr int KS = O::KS, UN = O::UN;                          // k per fragment step, steps of loads in flight
    constexpr int KG = KS / 4;                                     // k per lane group of a step
    __shared__ f32x4 red[NW - 1][SK_MF][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int nmf = (p.M + 15) >> 4;                               // 1..4 row fragments (uniform)
    const int kw = p.K / NW, k0 = wave * kw;                       // this wave's K range, a multiple of KS
    const TA* A = reinterpret_cast<const TA*>(p.a);
    // A block walks the 16-column tiles blockIdx.x, blockIdx.x + gridDim.x, ...: with the grid capped (sk_grid below) the
    // launch holds a handful of CUs instead of N / 16 of them for about as long - a tile is one memory round trip.
    for (int tile = blockIdx.x; tile < p.N / SK_COLS; tile += gridDim.x) {
    const int n0 = tile * SK_COLS;
    const TB* bp = reinterpret_cast<const TB*>(p.b) + (size_t)(n0 + col) * p.ldb + k0 + kg * KG;
    const TA* ap[SK_MF];
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) {
        int row = mf * 16 + col;
        row = row < p.M ? row : p.M - 1;                           // clamped rows are never stored
        ap[mf] = A + (size_t)row * p.lda + k0 + kg * KG;
    }
    f32x4 acc[SK_MF];
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < kw; ks += KS * UN) {
        typename O::braw bf[UN];
        typename O::afrag af[UN][SK_MF];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + KS * u < kw) {
                if constexpr (FFM_SKINNY_ABL & 1) bf[u] = {}; else
                O::loadB(bf[u], bp + ks + KS * u);
#pragma unroll
                for (int mf = 0; mf < SK_MF; ++mf) {
                    if constexpr (FFM_SKINNY_ABL & 2) af[u][mf] = {}; else
                    if (mf < nmf) O::loadA(af[u][mf], ap[mf] + ks + KS * u);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + KS * u < kw) {
                const typename O::bprep b = O::prep(bf[u]);
#pragma unroll
                for (int mf = 0; mf < SK_MF; ++mf)
                    if (mf < nmf) O::mma(acc[mf], af[u][mf], b);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mf = 0; mf < SK_MF; ++mf)
            if (mf < nmf) red[wave - 1][mf][lane] = acc[mf];
    }
    __syncthreads();
    if (wave == 0) {
    const int n = n0 + col;
    const float bias = (FL & FFM_EPI_BIAS) ? p.bias[n] : 0.f;
    TA* C = reinterpret_cast<TA*>(p.c);
#pragma unroll
    for (int mf = 0; mf < SK_MF; ++mf) {
        if (mf >= nmf) break;
        f32x4 v = acc[mf];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) {
            const f32x4 o = red[w][mf][lane];
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = mf * 16 + 4 * kg + e;
            if (row >= p.M) continue;
            const size_t o = (size_t)row * p.ldc + n;
            float x = v[e] + bias;
            if (FL & FFM_EPI_RESIDUAL) x += (float)reinterpret_cast<const TA*>(p.res)[o];
            if (FL & FFM_EPI_DGELU) {
                const float ax = (float)reinterpret_cast<const TA*>(p.aux)[o];
                x *= p.gelu_deriv ? ax : Act<TA>::gelu_grad(ax);
            }
            if ((FL & FFM_EPI_GELU) && p.gelu_deriv) {
                float ga, gd;
                Act<TA>::gelu_both(x, ga, gd);
                C[o] = (TA)gd;
                reinterpret_cast<TA*>(p.c2)[o] = (TA)ga;
            } else {
                C[o] = (TA)x;
                if (FL & FFM_EPI_GELU) reinterpret_cast<TA*>(p.c2)[o] = (TA)Act<TA>::gelu(x);
            }
        }
    }
    }
    if (tile + (int)gridDim.x < p.N / SK_COLS) __syncthreads();   // `red` is rewritten by the next tile
    }
}

// The X3 product with NT 16-column tiles per block.  Every block reads ALL rows of A (40 x K floats through the L2 -> CU
// path: 2.5x the bytes of its 16 weight rows), so with one tile per block the activations are most of a launch's traffic - and
// that traffic, not the CUs a launch occupies, is what the vision chain's kernels feel beside it (tools/side_proxy.py,
// tools/side_abl.sh).  The NT tiles of a block share the A fragments AND their hi / lo split; wave
// t < NT finishes tile t, and the partial sums meet in the one-tile kernel's order (wave 0, 1, ...): bit-identical results.
// MFC: row fragments (3 for M <= 48: the text tower's 40 rows), UN: K32 steps of loads in flight (2 when a wave's K slice is
// two steps, K = 512 on 8 waves).
template <typename TB, int NW, int FL, int NT, int MFC, int UN>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_nt_kernel(ffm_gemm_args p) {
    typedef SkOps<float, TB, true> O;
    constexpr int KS = O::KS, KG = KS / 4;
    static_assert(NT <= 8 && NW >= 4, "at most four tiles are finished at a time, one wave each");
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);                // [NW][min(NT, 4)][MFC][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int nmf = (p.M + 15) >> 4;
    const int kw = p.K / NW, k0 = wave * kw;
    const float* A = reinterpret_cast<const float*>(p.a);
    const int groups = p.N / (SK_COLS * NT);
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int n0 = grp * SK_COLS * NT;
    const TB* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = reinterpret_cast<const TB*>(p.b) + (size_t)(n0 + t * SK_COLS + col) * p.ldb + k0 + kg * KG;
    const float* ap[MFC];
#pragma unroll
    for (int mf = 0; mf < MFC; ++mf) {
        int row = mf * 16 + col;
        row = row < p.M ? row : p.M - 1;                           // clamped rows are never stored
        ap[mf] = A + (size_t)row * p.lda + k0 + kg * KG;
    }
    f32x4 acc[NT][MFC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mf = 0; mf < MFC; ++mf) acc[t][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < kw; ks += KS * UN) {
        typename O::braw bf[UN][NT];
        typename O::afrag af[UN][MFC];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + KS * u < kw) {
#pragma unroll
                for (int t = 0; t < NT; ++t) O::loadB(bf[u][t], bp[t] + ks + KS * u);
#pragma unroll
                for (int mf = 0; mf < MFC; ++mf)
                    if (mf < nmf) O::loadA(af[u][mf], ap[mf] + ks + KS * u);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ks + KS * u < kw) {
                typename O::bprep b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = O::prep(bf[u][t]);
#pragma unroll
                for (int mf = 0; mf < MFC; ++mf)
                    if (mf < nmf) {
                        const typename O::bprep a = O::prep(af[u][mf]);
#pragma unroll
                        for (int t = 0; t < NT; ++t) O::mma_split(acc[t][mf], a, b[t]);
                    }
            }
        }
    }
    // the partial sums meet in LDS, TG tiles at a time (8 tiles per block: two rounds through the same 96 KiB)
    constexpr int TG = NT > 4 ? 4 : NT;
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TG) {
    if (t0) __syncthreads();
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int mf = 0; mf < MFC; ++mf)
            if (mf < nmf) red[((wave * TG + t) * MFC + mf) * 64 + lane] = acc[t0 + t][mf];
    __syncthreads();
    if (wave < TG) {
        const int t = wave;
        const int n = n0 + (t0 + t) * SK_COLS + col;
        const float bias = (FL & FFM_EPI_BIAS) ? p.bias[n] : 0.f;
        float* C = reinterpret_cast<float*>(p.c);
        for (int mf = 0; mf < nmf; ++mf) {
            f32x4 v = red[((0 * TG + t) * MFC + mf) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const f32x4 o = red[((w * TG + t) * MFC + mf) * 64 + lane];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = mf * 16 + 4 * kg + e;
                if (row >= p.M) continue;
                const size_t o = (size_t)row * p.ldc + n;
                float x = v[e] + bias;
                if (FL & FFM_EPI_RESIDUAL) x += reinterpret_cast<const float*>(p.res)[o];
                if (FL & FFM_EPI_DGELU) {
                    const float ax = reinterpret_cast<const float*>(p.aux)[o];
                    x *= p.gelu_deriv ? ax : Act<float>::gelu_grad(ax);
                }
                if ((FL & FFM_EPI_GELU) && p.gelu_deriv) {
                    float ga, gd;
                    Act<float>::gelu_both(x, ga, gd);
                    C[o] = gd;
                    reinterpret_cast<float*>(p.c2)[o] = ga;
                } else {
                    C[o] = x;
                    if (FL & FFM_EPI_GELU) reinterpret_cast<float*>(p.c2)[o] = Act<float>::gelu(x);
                }
            }
        }
    }
    }
    if (grp + (int)gridDim.x < groups) __syncthreads();           // `red` is rewritten by the next tile group
    }
}


// ---- split over K across the grid (ABI 12, ffm_gemm_args.sk_part) ------------------------------------------------------------
// The text tower's narrow products (N = 512: c_proj forward, dX(c_fc), dX(qkv)) give the kernels above 16 column groups, i.e.
// 16 blocks that each walk K = 1536 / 2048 in two or three dependent memory round trips with 576 KB through ONE CU's L2 port:
// 20-28 us per launch, three dozen launches per step on the side stream - and all of the tower's products run at 8 waves x
// 150-240 registers, the footprint that keeps a CU from the vision chain's next panel launch (docs/experiments.md E4.6).  Here the grid is (column groups) x (K
// slices of 128): every block is 4 waves of ONE K32 step each - a single round trip, ~100 registers - and writes its 40 x 32
// partial tile to scratch [slice][M][N]; a second small launch sums the slices IN ORDER (deterministic: no atomics) and
// applies the epilogue.  More launches, but each is a few microseconds on many CUs with a footprint that fits beside a panel.
constexpr int SKS_K = 128;                                         // K per block (4 waves x one K32 step)
template <typename TB, int NT, int MFC>
__global__ __launch_bounds__(256) void gemm_skinny_splitk_kernel(ffm_gemm_args p) {
    typedef SkOps<float, TB, true> O;
    constexpr int KS = O::KS, KG = KS / 4;
    __shared__ f32x4 red[4][NT][MFC][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, kg = lane >> 4;
    const int nmf = (p.M + 15) >> 4;
    const int n0 = blockIdx.x * SK_COLS * NT, slice = blockIdx.y;
    const int k0 = slice * SKS_K + wave * KS;
    const float* A = reinterpret_cast<const float*>(p.a);
    typename O::braw bf[NT];
    typename O::afrag af[MFC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        O::loadB(bf[t], reinterpret_cast<const TB*>(p.b) + (size_t)(n0 + t * SK_COLS + col) * p.ldb + k0 + kg * KG);
#pragma unroll
    for (int mf = 0; mf < MFC; ++mf) {
        int row = mf * 16 + col;
        row = row < p.M ? row : p.M - 1;                           // clamped rows are never stored
        if (mf < nmf) O::loadA(af[mf], A + (size_t)row * p.lda + k0 + kg * KG);
    }
    f32x4 acc[NT][MFC];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mf = 0; mf < MFC; ++mf) acc[t][mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typename O::bprep b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = O::prep(bf[t]);
#pragma unroll
    for (int mf = 0; mf < MFC; ++mf)
        if (mf < nmf) {
            const typename O::bprep a = O::prep(af[mf]);
#pragma unroll
            for (int t = 0; t < NT; ++t) O::mma_split(acc[t][mf], a, b[t]);
        }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mf = 0; mf < MFC; ++mf)
            if (mf < nmf) red[wave][t][mf][lane] = acc[t][mf];
    __syncthreads();
    // the four K32 steps meet in wave order; wave t < NT finishes tile t, waves NT.. the remaining row fragments
    for (int item = wave; item < NT * nmf; item += 4) {
        const int t = item % NT, mf = item / NT;
        f32x4 v = red[0][t][mf][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 o = red[w][t][mf][lane];
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        const int n = n0 + t * SK_COLS + col;
        float* part = p.sk_part + (size_t)slice * p.M * p.N;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = mf * 16 + 4 * kg + e;
            if (row < p.M) part[(size_t)row * p.N + n] = v[e];
        }
    }
}

// sum of the K slices in slice order + the one-launch kernel's epilogue, four columns per thread
template <int FL>
__global__ __launch_bounds__(256) void skinny_splitk_finish_kernel(ffm_gemm_args p, int slices) {
    const int n4 = p.N >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * n4) return;
    const int row = idx / n4, n = (idx - row * n4) << 2;
    const size_t mn = (size_t)p.M * p.N;
    const float* part = p.sk_part + (size_t)row * p.N + n;
    f32x4 v = *reinterpret_cast<const f32x4*>(part);
    for (int s = 1; s < slices; ++s) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(part + (size_t)s * mn);
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    const size_t o = (size_t)row * p.ldc + n;
    float* C = reinterpret_cast<float*>(p.c);
    f32x4 out, out2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + ((FL & FFM_EPI_BIAS) ? p.bias[n + e] : 0.f);
        if (FL & FFM_EPI_RESIDUAL) x += reinterpret_cast<const float*>(p.res)[o + e];
        if (FL & FFM_EPI_DGELU) {
            const float ax = reinterpret_cast<const float*>(p.aux)[o + e];
            x *= p.gelu_deriv ? ax : Act<float>::gelu_grad(ax);
        }
        if ((FL & FFM_EPI_GELU) && p.gelu_deriv) {
            float ga, gd;
            Act<float>::gelu_both(x, ga, gd);
            out[e] = gd;
            out2[e] = ga;
        } else {
            out[e] = x;
            if (FL & FFM_EPI_GELU) out2[e] = Act<float>::gelu(x);
        }
    }
    *reinterpret_cast<f32x4*>(C + o) = out;
    if (FL & FFM_EPI_GELU) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c2) + o) = out2;
}

// which products are split: at most 48 rows, N up to FFM_SKINNY_SPLITK (default 2048: every product of the text tower; 0
// switches it off: A/B runs), K of at least FFM_SKINNY_SPLITK_MIN (default 4) slices of 128.  Measured on the bench step, three
// alternating runs each in one call (profiles/r06_splitk_ab3.txt): off 4.669, N <= 512 only 4.677, all products 4.652 ms
inline int sk_slices(int M, int N, int K) {
    static const int nmax = getenv("FFM_SKINNY_SPLITK") ? atoi(getenv("FFM_SKINNY_SPLITK")) : 2048;
    static const int smin = getenv("FFM_SKINNY_SPLITK_MIN") ? atoi(getenv("FFM_SKINNY_SPLITK_MIN")) : 4;
    if (M > 48 || N > nmax || N % 32 || K % SKS_K || K / SKS_K < smin || K / SKS_K > 64) return 0;
    return K / SKS_K;
}

template <typename TB>
int launch_splitk(const ffm_gemm_args& a, hipStream_t s) {
    const int slices = sk_slices(a.M, a.N, a.K);
    if ((a.ldc & 3) || (a.N & 3)) return FFM_EUNSUP;
    hipLaunchKernelGGL((gemm_skinny_splitk_kernel<TB, 2, 3>), dim3(a.N / (SK_COLS * 2), slices), dim3(256), 0, s, a);
    FFM_CHECK_LAUNCH();
    const dim3 g((a.M * (a.N >> 2) + 255) / 256);
    switch (a.flags) {
        case 0: hipLaunchKernelGGL((skinny_splitk_finish_kernel<0>), g, dim3(256), 0, s, a, slices); break;
        case FFM_EPI_BIAS: hipLaunchKernelGGL((skinny_splitk_finish_kernel<FFM_EPI_BIAS>), g, dim3(256), 0, s, a, slices); break;
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL: hipLaunchKernelGGL((skinny_splitk_finish_kernel<FFM_EPI_BIAS | FFM_EPI_RESIDUAL>), g, dim3(256), 0, s, a, slices); break;
        case FFM_EPI_BIAS | FFM_EPI_GELU: hipLaunchKernelGGL((skinny_splitk_finish_kernel<FFM_EPI_BIAS | FFM_EPI_GELU>), g, dim3(256), 0, s, a, slices); break;
        case FFM_EPI_DGELU: hipLaunchKernelGGL((skinny_splitk_finish_kernel<FFM_EPI_DGELU>), g, dim3(256), 0, s, a, slices); break;
        default: return FFM_EINVAL;
    }
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

// Blocks per launch.  Every block of the vision tower's single-round panel GEMMs needs a whole CU, and a panel launch leaves
// 8-16 of the 256 idle: a side-stream launch that holds more CUs than that when a panel starts keeps some of its blocks
// waiting (the side streams cost the chain ~0.4 ms per step).  FFM_SKINNY_CAP=<n> caps the grid at n
// blocks (0 = one block per tile).
inline int sk_grid(int tiles) {
    static const int cap = getenv("FFM_SKINNY_CAP") ? atoi(getenv("FFM_SKINNY_CAP")) : FFM_SKINNY_CAP_DEFAULT;
    return cap > 0 && tiles > cap ? cap : tiles;
}

// Tiles per block of the X3 product: FFM_SKINNY_NT = 1 | 2 | 4; a product whose N does not divide takes fewer.
inline int sk_nt(int N) {
    static const int want = getenv("FFM_SKINNY_NT") ? atoi(getenv("FFM_SKINNY_NT")) : FFM_SKINNY_NT_DEFAULT;
    static const int minb = getenv("FFM_SKINNY_MINB") ? atoi(getenv("FFM_SKINNY_MINB")) : FFM_SKINNY_MINB_DEFAULT;
    static const int want_n = getenv("FFM_SKINNY_NT_NARROW") ? atoi(getenv("FFM_SKINNY_NT_NARROW")) : want;   // N <= 512
    const int w = N <= 512 ? want_n : want;
    int nt = w >= 4 ? 4 : (w >= 2 ? 2 : 1);
    while (nt > 1 && (N % (SK_COLS * nt) || N / (SK_COLS * nt) < minb)) nt >>= 1;     // never fewer than minb blocks
    return nt;
}

template <typename TB, int NW, int FL, int NT, int MFC, int UN>
int launch_nt(const ffm_gemm_args& a, hipStream_t s) {
    constexpr int lds = NW * (NT > 4 ? 4 : NT) * MFC * 64 * 16;
    static bool done = false;                         // one per instantiation
    if (!done && lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_nt_kernel<TB, NW, FL, NT, MFC, UN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return (int)e;
    }
    done = true;
    hipLaunchKernelGGL((gemm_skinny_nt_kernel<TB, NW, FL, NT, MFC, UN>), dim3(sk_grid(a.N / (SK_COLS * NT))), dim3(NW * 64), lds, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

template <typename TA, typename TB, bool X3, int NW, int FL>
int launch(const ffm_gemm_args& a, hipStream_t s) {
    if constexpr (X3 && NW == 8) {
        // (the text tower's shapes: 40 rows, K = 512 or 2048 on 8 waves -> 2 or 8 K32 steps per wave)
        const int nt = a.M <= 48 ? sk_nt(a.N) : 1, steps = a.K / NW / 32;
        // (eight tiles per block for the wide products - 16 blocks, two rounds through the meeting buffer - measured: 4.66
        // against 4.57 ms per step; not instantiated)
        if (nt == 4 && steps <= 2) return launch_nt<TB, NW, FL, 4, 3, 2>(a, s);
        if (nt >= 2 && steps <= 2) return launch_nt<TB, NW, FL, 2, 3, 2>(a, s);
        if (nt >= 2) return launch_nt<TB, NW, FL, 2, 3, 4>(a, s);
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<TA, TB, X3, NW, FL>), dim3(sk_grid(a.N / SK_COLS)), dim3(NW * 64), 0, s, a);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

template <typename TA, typename TB, bool X3, int NW>
int launch_flags(const ffm_gemm_args& a, hipStream_t s) {
    switch (a.flags) {
        case 0: return launch<TA, TB, X3, NW, 0>(a, s);
        case FFM_EPI_BIAS: return launch<TA, TB, X3, NW, FFM_EPI_BIAS>(a, s);
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL: return launch<TA, TB, X3, NW, FFM_EPI_BIAS | FFM_EPI_RESIDUAL>(a, s);
        case FFM_EPI_BIAS | FFM_EPI_GELU: return launch<TA, TB, X3, NW, FFM_EPI_BIAS | FFM_EPI_GELU>(a, s);
        case FFM_EPI_DGELU: return launch<TA, TB, X3, NW, FFM_EPI_DGELU>(a, s);
    }
    return FFM_EINVAL;
}

// 8 waves split K when it divides (one round of loads per wave for K <= 2048), 4 otherwise
template <typename TA, typename TB, bool X3>
int launch_waves(const ffm_gemm_args& a, hipStream_t s) {
    return a.K % 256 == 0 ? launch_flags<TA, TB, X3, 8>(a, s) : launch_flags<TA, TB, X3, 4>(a, s);
}

}  // namespace

bool ffm_skinny_ok(const ffm_gemm_args& a, int dtype) {
    if ((dtype != FFM_BF16 && dtype != FFM_F32 && dtype != FFM_F32_X3 && dtype != FFM_F32_X3_W16) || a.M > 16 * SK_MF || a.N % SK_COLS || a.K % 128)
        return false;
    switch (a.flags) {
        case 0:
        case FFM_EPI_BIAS:
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL:
        case FFM_EPI_BIAS | FFM_EPI_GELU:
        case FFM_EPI_DGELU: return true;
    }
    return false;
}

#ifndef FFM_TWIN_F16     // (defined once: the split products are float32-storage only, the half twin of this file has no use for it)
extern "C" int64_t ffm_gemm_splitk_floats(int M, int N, int K, int dtype) {
    if (dtype != FFM_F32_X3 && dtype != FFM_F32_X3_W16) return 0;
    return (int64_t)sk_slices(M, N, K) * M * N;
}
#endif

int ffm_skinny_launch(const ffm_gemm_args& a, int dtype, hipStream_t s) {
    if (a.sk_part && (dtype == FFM_F32_X3 || dtype == FFM_F32_X3_W16) && sk_slices(a.M, a.N, a.K) > 0 && !(a.ldc & 3)) {
        const int e = dtype == FFM_F32_X3 ? launch_splitk<float>(a, s) : launch_splitk<_Float16>(a, s);
        if (e != FFM_EUNSUP) return e;
    }
    if (dtype == FFM_F32) return launch_waves<float, float, false>(a, s);
    if (dtype == FFM_F32_X3) return launch_waves<float, float, true>(a, s);
    if (dtype == FFM_F32_X3_W16) return launch_waves<float, _Float16, true>(a, s);
    return launch_waves<bf16_t, bf16_t, false>(a, s);
}
