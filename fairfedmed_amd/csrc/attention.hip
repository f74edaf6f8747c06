// Multi-head self-attention, head_dim 64, whole key range per workgroup
// (L <= 256: 197 vision tokens, 77 text tokens), MFMA 16x16.
//
// Orientation: scores are computed transposed, S^T = K Q^T, so a lane holds one
// query column (q = lane & 15) and 4 consecutive keys per 16-key fragment
// (key = 16 f + 4 (lane>>4) + e).  That accumulator is directly the B operand of
// the next product (O^T = V^T P^T, dQ^T = K^T dS^T, ...) with the k-order
// permuted identically on the A side, so P never goes through LDS.  The A side
// of those products needs the key (or query) index contiguous per lane, so V /
// K / Q / dO are staged TRANSPOSED in LDS ([64 d][tokens], zero padded).
//
//   forward : one block per (b, h, q-split); waves own 16-row q tiles.
//   backward: dQ kernel (same shape as forward); delta = rowsum(dO * O) is formed inside both kernels;
//             dK/dV kernel: waves own 16-key tiles and sweep all queries.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HD = 64;

template <typename T> struct AT;     // per-dtype attention helpers
template <> struct AT<bf16_t> {
    typedef bf16x8 frag_t;
    static constexpr int ES = 2, ROWB = 128, NCH = 8, ND = 2, CE = 8;
    static constexpr int FPK = 2;    // 16-token fragments per second-product k-step
    static __device__ __forceinline__ frag_t pack(const f32x4* p, int s) {
        const f32x4 a = p[2 * s], b = p[2 * s + 1];
        frag_t r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3],
                    (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
        return r;
    }
    // A fragment of a transposed tile Xt[d][token] for k-step s: tokens
    // 32 s + 16 (j>>2) + 4 g + (j&3), j = 0..7  (matches pack()).
    static __device__ __forceinline__ frag_t tfrag(const bf16_t* xt, int ts, int d, int s, int g) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(xt + (size_t)d * ts + 32 * s + 4 * g);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(xt + (size_t)d * ts + 32 * s + 16 + 4 * g);
        frag_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
    }
};
template <> struct AT<float> {
    typedef f32x4 frag_t;
    static constexpr int ES = 4, ROWB = 256, NCH = 16, ND = 4, CE = 4;
    static constexpr int FPK = 1;
    static __device__ __forceinline__ frag_t pack(const f32x4* p, int s) { return p[s]; }
    static __device__ __forceinline__ frag_t tfrag(const float* xt, int ts, int d, int s, int g) {
        return *reinterpret_cast<const f32x4*>(xt + (size_t)d * ts + 16 * s + 4 * g);
    }
};

// element stride of a transposed tile row: bytes = roundup(LP*ES, 32) + 16
// (rows 16 B apart modulo 32 B keep the 16-row fragment reads off each other's banks)
template <typename T> __host__ __device__ inline int tstride(int LP) {
    const int bytes = ((LP * AT<T>::ES + 31) / 32) * 32 + 16;
    return bytes / AT<T>::ES;
}

// swizzled byte offset of 16-B chunk `c` of row `row` in a row-major [rows][64] LDS tile
template <typename T> __device__ __forceinline__ int rm_off(int row, int c) {
    return row * AT<T>::ROWB + ((c ^ (row & (AT<T>::NCH - 1))) << 4);
}

// row-major tile (zero rows >= L) -> LDS, swizzled.  All global loads are issued before the
// first LDS write (one memory round trip per tile instead of one per loop iteration).
template <typename T, int LP, int NT>
__device__ __forceinline__ void stage_rowmajor(const T* __restrict__ src, int ld, int L, char* dst, int tid) {
    typedef typename AT<T>::frag_t frag_t;
    constexpr int TOTAL = LP * AT<T>::NCH, NIT = (TOTAL + NT - 1) / NT;
    frag_t buf[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int row = idx / AT<T>::NCH, c = idx % AT<T>::NCH;
#pragma unroll
        for (int e = 0; e < AT<T>::CE; ++e) buf[it][e] = (T)0.f;
        if (idx < TOTAL && row < L) buf[it] = *reinterpret_cast<const frag_t*>(src + (size_t)row * ld + c * AT<T>::CE);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int row = idx / AT<T>::NCH, c = idx % AT<T>::NCH;
        if (idx < TOTAL) *reinterpret_cast<frag_t*>(dst + rm_off<T>(row, c)) = buf[it];
    }
}

// transposed tile Xt[d][token] (zero tokens >= L) -> LDS
template <typename T, int LP, int NT>
__device__ __forceinline__ void stage_transposed(const T* __restrict__ src, int ld, int L, T* dst, int ts, int tid) {
    typedef typename AT<T>::frag_t frag_t;
    // consecutive threads take consecutive tokens (conflict-free LDS writes), chunks outer
    constexpr int TOTAL = LP * AT<T>::NCH, NIT = (TOTAL + NT - 1) / NT;
    frag_t buf[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int c = idx / LP, tok = idx % LP;
#pragma unroll
        for (int e = 0; e < AT<T>::CE; ++e) buf[it][e] = (T)0.f;
        if (idx < TOTAL && tok < L) buf[it] = *reinterpret_cast<const frag_t*>(src + (size_t)tok * ld + c * AT<T>::CE);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int c = idx / LP, tok = idx % LP;
        if (idx < TOTAL) {
#pragma unroll
            for (int e = 0; e < AT<T>::CE; ++e) dst[(size_t)(c * AT<T>::CE + e) * ts + tok] = buf[it][e];
        }
    }
}

// 16-B operand fragment of row `row` (clamped) straight from global memory
template <typename T>
__device__ __forceinline__ typename AT<T>::frag_t gfrag(const T* __restrict__ src, int ld, int row, int L, int ks,
                                                        int g) {
    const int r = row < L ? row : L - 1;
    return *reinterpret_cast<const typename AT<T>::frag_t*>(src + (size_t)r * ld + (ks * 4 + g) * AT<T>::CE);
}

__device__ __forceinline__ float group4_max(float v) {   // lanes l, l^16, l^32, l^48 share a column
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename T, int NFP, int NTH>
__global__ __launch_bounds__(NTH) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                       float* __restrict__ lse, int L, int heads, int causal,
                                                       int qsplit) {
    typedef typename AT<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int LP = NFP * 16;
    const int ts = tstride<T>(LP);
    char* Ks = smem;                                              // [LP][64] swizzled
    T* Vt = reinterpret_cast<T*>(smem + LP * AT<T>::ROWB);        // [64][ts]

    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + h * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;

    stage_rowmajor<T, LP, NTH>(base + E, ld, L, Ks, tid);
    stage_transposed<T, LP, NTH>(base + 2 * E, ld, L, Vt, ts, tid);
    __syncthreads();

    const int NF = (L + 15) / 16;
    const int per = (NF + qsplit - 1) / qsplit;
    const int qt0 = blockIdx.y * per;
    const int qt1 = (qt0 + per) < NF ? (qt0 + per) : NF;
    for (int qt = qt0 + wave; qt < qt1; qt += NTH / 64) {
        const int q = qt * 16 + col;
        frag_t qf[AT<T>::ND];
#pragma unroll
        for (int ks = 0; ks < AT<T>::ND; ++ks) qf[ks] = gfrag<T>(base, ld, q, L, ks, g);

        f32x4 s[NFP];
        float mx = -INFINITY;
        frag_t kcur[AT<T>::ND];
#pragma unroll
        for (int ks = 0; ks < AT<T>::ND; ++ks)
            kcur[ks] = *reinterpret_cast<const frag_t*>(Ks + rm_off<T>(col, ks * 4 + g));
#pragma unroll
        for (int f = 0; f < NFP; ++f) {
            frag_t knxt[AT<T>::ND];                   // fragments of f+1 are in flight while f multiplies
#pragma unroll
            for (int ks = 0; ks < AT<T>::ND; ++ks)
                knxt[ks] = *reinterpret_cast<const frag_t*>(Ks + rm_off<T>((f + 1 < NFP ? f + 1 : f) * 16 + col, ks * 4 + g));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < AT<T>::ND; ++ks) Mma16<T>::mma(acc, kcur[ks], qf[ks]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = f * 16 + g * 4 + e;
                float v = acc[e] * 0.125f;
                if (key >= L || (causal && key > q)) v = -INFINITY;
                acc[e] = v;
                mx = fmaxf(mx, v);
            }
            s[f] = acc;
#pragma unroll
            for (int ks = 0; ks < AT<T>::ND; ++ks) kcur[ks] = knxt[ks];
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment loads from being hoisted further (VGPR pressure)
        }
        mx = group4_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int f = 0; f < NFP; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = fast_expf(s[f][e] - mx);
                s[f][e] = p;
                sum += p;
            }
        sum = group4_sum(sum);

        f32x4 o[4];
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) o[fd] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int NST = NFP / AT<T>::FPK;
        frag_t vcur[4];
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) vcur[fd] = AT<T>::tfrag(Vt, ts, fd * 16 + col, 0, g);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            frag_t vnxt[4];
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) vnxt[fd] = AT<T>::tfrag(Vt, ts, fd * 16 + col, st + 1 < NST ? st + 1 : st, g);
            const frag_t pf = AT<T>::pack(s, st);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) Mma16<T>::mma(o[fd], vcur[fd], pf);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) vcur[fd] = vnxt[fd];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < L) {
            const float inv = 1.0f / sum;
            T* orow = out + ((size_t)b * L + q) * E + h * HD;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                f32x4 v = o[fd];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= inv;
                Vec4<T>::store(orow + fd * 16 + g * 4, v);
            }
            if (g == 0 && lse) lse[((size_t)b * heads + h) * L + q] = mx + __logf(sum);
        }
    }
}

// Backward kernels: one block per (b, h), one wave per 16-token tile (bf16; 4 waves in f32 mode).  With STAGED (bf16) the row-major
// operand tiles live in LDS next to the transposed ones; the f32 parity mode would need > 160 KB
// for that and reads those fragments from global memory (L2) instead.

template <typename T, bool STAGED>
__device__ __forceinline__ typename AT<T>::frag_t opfrag(const char* lds_tile, const T* __restrict__ src, int ld,
                                                        int row, int L, int ks, int g) {
    if constexpr (STAGED)
        return *reinterpret_cast<const typename AT<T>::frag_t*>(lds_tile + rm_off<T>(row, ks * 4 + g));
    else
        return gfrag<T>(src, ld, row, L, ks, g);
}

// dQ: dS^T[key][q] = P^T * (dP^T - delta[q]) * scale,  dQ^T[d][q] = sum_key Kt[d][key] dS^T[key][q].
template <typename T, int NFP, bool STAGED, int BW_THREADS>
__global__ __launch_bounds__(BW_THREADS) void attn_bwd_dq_kernel(const T* __restrict__ qkv,
                                                                 const T* __restrict__ d_o,
                                                                 const float* __restrict__ lse,
                                                                 const T* __restrict__ o_fwd,
                                                                 T* __restrict__ dqkv, int L, int heads, int causal) {
    typedef typename AT<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int LP = NFP * 16;
    const int ts = tstride<T>(LP);
    T* Kt = reinterpret_cast<T*>(smem);                           // [64][ts]
    char* Ks = smem + (size_t)HD * ts * AT<T>::ES;                // [LP][64] swizzled (STAGED)
    char* Vs = Ks + LP * AT<T>::ROWB;

    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + h * HD;
    const T* dob = d_o + (size_t)b * L * E + h * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;

    stage_transposed<T, LP, BW_THREADS>(base + E, ld, L, Kt, ts, tid);
    if constexpr (STAGED) {
        stage_rowmajor<T, LP, BW_THREADS>(base + E, ld, L, Ks, tid);
        stage_rowmajor<T, LP, BW_THREADS>(base + 2 * E, ld, L, Vs, tid);
    }
    __syncthreads();

    const int NF = (L + 15) / 16;
    for (int qt = wave; qt < NF; qt += BW_THREADS / 64) {
        // (f32 path) the K/V fragment loads below do not depend on qt: without this compiler-level
        // fence LICM hoists all of them out of the loop and keeps NFP*ND*2 fragments live (spills)
        asm volatile("" ::: "memory");
        const int q = qt * 16 + col;
        const int qc = q < L ? q : L - 1;
        frag_t qf[AT<T>::ND], dof[AT<T>::ND];
#pragma unroll
        for (int ks = 0; ks < AT<T>::ND; ++ks) {
            qf[ks] = gfrag<T>(base, ld, q, L, ks, g);
            dof[ks] = gfrag<T>(dob, E, q, L, ks, g);
        }
        const float lq = lse[((size_t)b * heads + h) * L + qc];
        // delta[q] = sum_d dO[q][d] O[q][d]: the lane already holds 16 of the 64 dO values of its query; the four lane
        // groups of the column cover the row (this replaced a separate kernel and its launch)
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < AT<T>::ND; ++ks) {
            const frag_t of = gfrag<T>(o_fwd + (size_t)b * L * E + h * HD, E, q, L, ks, g);
#pragma unroll
            for (int e = 0; e < AT<T>::CE; ++e) dl += (float)of[e] * (float)dof[ks][e];
        }
        dl = group4_sum(dl);

        // dS^T of one k-step (FPK key fragments) is formed and consumed right away (no 14-fragment array, no software
        // prefetch: the other waves of the SIMD cover the LDS round trips, the registers saved keep 3-4 of them there)
        f32x4 dq[4];
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) dq[fd] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int FPK = AT<T>::FPK, NST = NFP / FPK;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x4 d2[FPK];
#pragma unroll
            for (int ff = 0; ff < FPK; ++ff) {
                const int f = st * FPK + ff;
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < AT<T>::ND; ++ks) {
                    const frag_t kc = opfrag<T, STAGED>(Ks, base + E, ld, f * 16 + col, L, ks, g);
                    const frag_t vc = opfrag<T, STAGED>(Vs, base + 2 * E, ld, f * 16 + col, L, ks, g);
                    Mma16<T>::mma(sa, kc, qf[ks]);
                    Mma16<T>::mma(pa, vc, dof[ks]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = f * 16 + g * 4 + e;
                    float v = 0.f;
                    if (key < L && !(causal && key > q)) {
                        const float p = fast_expf(sa[e] * 0.125f - lq);
                        v = p * (pa[e] - dl) * 0.125f;
                    }
                    sa[e] = v;
                }
                d2[ff] = sa;
            }
            const frag_t df = AT<T>::pack(d2, 0);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) Mma16<T>::mma(dq[fd], AT<T>::tfrag(Kt, ts, fd * 16 + col, st, g), df);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < L) {
            T* drow = dqkv + ((size_t)b * L + q) * ld + h * HD;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) Vec4<T>::store(drow + fd * 16 + g * 4, dq[fd]);
        }
    }
}

// dK/dV: waves own 16-key tiles; a lane holds key = lane&15 and 4 consecutive
// queries per fragment.  S[q][key] = Q K^T (A = Q rows), dP[q][key] = dO V^T,
// dV^T[d][key] = sum_q dOt[d][q] P[q][key],  dK^T[d][key] = sum_q Qt[d][q] dS[q][key].
template <typename T, int NFP, bool STAGED, int BW_THREADS>
__global__ __launch_bounds__(BW_THREADS) void attn_bwd_dkv_kernel(const T* __restrict__ qkv,
                                                                  const T* __restrict__ d_o,
                                                                  const float* __restrict__ lse,
                                                                  const T* __restrict__ o_fwd,
                                                                  T* __restrict__ dqkv, int L, int heads, int causal) {
    typedef typename AT<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int LP = NFP * 16;
    const int ts = tstride<T>(LP);
    T* Qt = reinterpret_cast<T*>(smem);                           // [64][ts]
    T* dOt = Qt + (size_t)HD * ts;                                // [64][ts]
    float* lse_s = reinterpret_cast<float*>(dOt + (size_t)HD * ts);   // [LP]
    float* del_s = lse_s + LP;                                    // [LP]
    char* Qs = reinterpret_cast<char*>(del_s + LP);               // [LP][64] swizzled (STAGED)
    char* dOs = Qs + LP * AT<T>::ROWB;

    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const T* base = qkv + (size_t)b * L * ld + h * HD;
    const T* dob = d_o + (size_t)b * L * E + h * HD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;

    stage_transposed<T, LP, BW_THREADS>(base, ld, L, Qt, ts, tid);
    stage_transposed<T, LP, BW_THREADS>(dob, E, L, dOt, ts, tid);
    if constexpr (STAGED) {
        stage_rowmajor<T, LP, BW_THREADS>(base, ld, L, Qs, tid);
        stage_rowmajor<T, LP, BW_THREADS>(dob, E, L, dOs, tid);
    }
    for (int i = tid; i < LP; i += BW_THREADS) lse_s[i] = i < L ? lse[((size_t)b * heads + h) * L + i] : 0.f;
    // delta[q] = sum_d dO[q][d] O[q][d], 4 threads per query row (16 values each), all loads issued before the sums
    {
        const T* ob = o_fwd + (size_t)b * L * E + h * HD;
        constexpr int NQ = (4 * LP + BW_THREADS - 1) / BW_THREADS, NC = 16 / AT<T>::CE;
        frag_t fo[NQ][NC], fd[NQ][NC];
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int idx = tid + it * BW_THREADS, row = idx >> 2, part = idx & 3;
            const int rr = row < L ? row : L - 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                fo[it][c] = *reinterpret_cast<const frag_t*>(ob + (size_t)rr * E + part * 16 + c * AT<T>::CE);
                fd[it][c] = *reinterpret_cast<const frag_t*>(dob + (size_t)rr * E + part * 16 + c * AT<T>::CE);
            }
        }
#pragma unroll
        for (int it = 0; it < NQ; ++it) {
            const int idx = tid + it * BW_THREADS, row = idx >> 2;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int e = 0; e < AT<T>::CE; ++e) acc += (float)fo[it][c][e] * (float)fd[it][c][e];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if ((idx & 3) == 0 && row < LP) del_s[row] = row < L ? acc : 0.f;
        }
    }
    __syncthreads();

    const int NF = (L + 15) / 16;
    for (int kt = wave; kt < NF; kt += BW_THREADS / 64) {
        asm volatile("" ::: "memory");   // same LICM fence as in the dQ kernel (Q / dO fragments)
        const int key = kt * 16 + col;
        frag_t kf[AT<T>::ND], vf[AT<T>::ND];
#pragma unroll
        for (int ks = 0; ks < AT<T>::ND; ++ks) {
            kf[ks] = gfrag<T>(base + E, ld, key, L, ks, g);
            vf[ks] = gfrag<T>(base + 2 * E, ld, key, L, ks, g);
        }
        // The two products of a k-step (FPK query fragments) are formed and consumed right away: P and dS of the whole
        // query range are never live together (that was 112 registers and kept this kernel at 2 waves per SIMD).
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) { dv[fd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[fd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        constexpr int FPK = AT<T>::FPK, NST = NFP / FPK;
        // (No software prefetch of the next fragments here: at 3-4 waves per SIMD the other waves cover the LDS
        // round trip, and the prefetch registers are exactly what would push the kernel over 128 VGPRs.)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x4 p2[FPK], d2[FPK];
#pragma unroll
            for (int ff = 0; ff < FPK; ++ff) {
                const int f = st * FPK + ff;
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < AT<T>::ND; ++ks) {
                    const frag_t qc = opfrag<T, STAGED>(Qs, base, ld, f * 16 + col, L, ks, g);
                    const frag_t dc = opfrag<T, STAGED>(dOs, dob, E, f * 16 + col, L, ks, g);
                    Mma16<T>::mma(sa, qc, kf[ks]);
                    Mma16<T>::mma(pa, dc, vf[ks]);
                }
                const int qb = f * 16 + g * 4;                    // this lane's 4 consecutive queries
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&lse_s[qb]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&del_s[qb]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = qb + e;
                    float p = 0.f, d = 0.f;
                    if (q < L && key < L && !(causal && key > q)) {
                        p = fast_expf(sa[e] * 0.125f - l4[e]);
                        d = p * (pa[e] - d4[e]) * 0.125f;
                    }
                    sa[e] = p;
                    pa[e] = d;
                }
                p2[ff] = sa;
                d2[ff] = pa;
            }
            const frag_t pf = AT<T>::pack(p2, 0);
            const frag_t df = AT<T>::pack(d2, 0);
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                Mma16<T>::mma(dv[fd], AT<T>::tfrag(dOt, ts, fd * 16 + col, st, g), pf);
                Mma16<T>::mma(dk[fd], AT<T>::tfrag(Qt, ts, fd * 16 + col, st, g), df);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (key < L) {
            T* drow = dqkv + ((size_t)b * L + key) * ld + h * HD;
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                Vec4<T>::store(drow + E + fd * 16 + g * 4, dk[fd]);
                Vec4<T>::store(drow + 2 * E + fd * 16 + g * 4, dv[fd]);
            }
        }
    }
}

template <typename T> constexpr bool kStaged() { return sizeof(T) == 2; }
// bf16: one wave per 16-token tile (a 16-row tile of dQ or a 16-key tile of dK/dV), up to 14 waves per block, so a
// wave never runs two tiles back to back (with 8 waves the 13 tiles of L = 197 took two rounds inside every block);
// longer sequences fall back to 8 waves.  f32 needs the 512-VGPR budget of 256 threads.
template <typename T, int NFP> constexpr int kBwThreads() { return sizeof(T) == 2 ? (NFP <= 14 ? (NFP < 8 ? 512 : 64 * NFP) : 512) : 256; }
template <typename T> int lds_fwd(int NFP) { return NFP * 16 * AT<T>::ROWB + HD * tstride<T>(NFP * 16) * AT<T>::ES; }
template <typename T> int lds_dq(int NFP) {
    return HD * tstride<T>(NFP * 16) * AT<T>::ES + (kStaged<T>() ? 2 * NFP * 16 * AT<T>::ROWB : 0);
}
template <typename T> int lds_dkv(int NFP) {
    return 2 * HD * tstride<T>(NFP * 16) * AT<T>::ES + 2 * NFP * 16 * 4 + (kStaged<T>() ? 2 * NFP * 16 * AT<T>::ROWB : 0);
}

template <typename F> int set_lds(F fn, int bytes) {
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Second-generation kernels for the vision tower (bf16, no mask, 8..14 key fragments: L = 113..224 tokens).
//
// What the first generation spends its time on (rocprofv3 + SQ counters, profiles/r01_v6_sq_counters.json: MFMA pipes
// 5-9 % busy): (i) one block per (b, h) = 384 blocks of 14 waves on 256 CUs, two per CU on half of the chip and one on
// the rest; (ii) ~10 VALU instructions per score element (scale, mask, max, subtract, exp, ...) at 6-7 waves per SIMD;
// (iii) V (and K / Q / dO in the backward kernels) staged TRANSPOSED through registers with 2-byte LDS stores.
// Here:
//   * a block is HALF a head's 16-row tiles (7 + 6 for L = 197): 768 blocks = exactly three per CU at 7 waves each,
//     the two halves of a head on one XCD (block ids b and b + 8), so the second fetch of K / V is an L2 hit;
//   * K and V land ROW-major in LDS by LDS-DMA (8 rows x 128 B per wave instruction, the XOR swizzle applied to the
//     source address); the transposed operand of O^T = V^T P^T comes from ds_read_b64_tr_b16, so nothing is staged
//     through registers and no tile exists twice;
//   * the scores stay unscaled: p = exp2(c s - c m) is one FMA and one exp per element (c = log2(e) / 8), the row
//     maximum is taken on the raw scores, and only the fragment that straddles L is masked;
//   * two key chunks with a running maximum (8 fragments, then the rest): 32 score registers instead of 56, which is
//     what lets three blocks share a CU (<= 80 registers per lane).
// Rows >= L of the LDS tiles are copies of row L - 1 (the DMA source is clamped): finite, and always multiplied by an
// exactly zero probability.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;
// Diagnostic builds only (tools/attn_phases.sh): -DFFM_ATTN_ABL=1 the attn2 kernels return behind their load phase,
// =2 they skip the tile DMA and compute on whatever LDS holds (results are garbage): prices the two phases.
#ifndef FFM_ATTN_ABL
#define FFM_ATTN_ABL 0
#endif

// 16-B chunk c of row `row` in a row-major [rows][64] bf16 tile (128-byte rows): the same image as rm_off<bf16_t>
__device__ __forceinline__ int rm2(int row, int c) { return row * 128 + ((c ^ (row & 7)) << 4); }

// rows [0, nrows) of the token-major matrix `src` (row stride ld elements, 64 columns) -> swizzled LDS tile, by DMA
// (nrows = L rounded up to 8: the tile's rows)
template <int NW>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ src, int ld, int L, int nrows, char* dst, int wave, int lane) {
    const int rsub = lane >> 3, slot = lane & 7;
    if constexpr ((FFM_ATTN_ABL & 2) != 0) return;
    for (int pc = wave; pc < nrows / 8; pc += NW) {
        int row = pc * 8 + rsub;
        const int chunk = slot ^ (row & 7);
        row = row < L ? row : L - 1;
        const bf16_t* g = src + (size_t)row * ld + chunk * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
    }
}

// A-operand fragment (8 bf16) of X^T for the k-step whose keys are 32 s + 16 (j >> 2) + 4 g + (j & 3), rows d = 16 fd + (lane & 15):
// two transposed reads of the row-major tile X [token][64] (cdna_hip_programming.md T10: lane 4q + p of a 16-lane
// group addresses row q, columns 4p .. 4p + 3 of a 4 x 16 block and receives column (lane & 15) of its four rows)
// Rows are clamped to the tile (rmax = its last row): a lane supplies its own address, and whatever a clamped row
// delivers meets an exactly zero probability.
// CLAMP (the peeled last k-step of a tile with fewer than 16 NF rows): only the last 8 rows of fragment NF - 1 and the
// phantom fragment of an odd NF can lie beyond such a tile.  Without it the rows past the tile are whatever follows it
// in LDS: callers put another bf16 tile there.
template <bool CLAMP>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int s, int fd, int lane, int rmax) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    int r0 = 32 * s + 4 * g + q, r1 = r0 + 16;
    if constexpr (CLAMP) {
        r0 = r0 < rmax ? r0 : rmax;
        r1 = r1 < rmax ? r1 : rmax;
    }
    const int c = fd * 2 + (p >> 1), half = (p & 1) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + rm2(r0, c) + half));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + rm2(r1, c) + half));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// tr_frag with the address split into a lane part the caller keeps in ONE register and cheap per-use arithmetic:
//   address(s, fd) = ((lane0 + 4096 s) ^ (fd << 5)) [+ 2048 for the second half]
// lane0 = tr_lane0(lane) is the address for s = 0, fd = 0: row 4 g + q, chunk (p >> 1) ^ (row & 7), + 8 bytes for odd p.
// A different fd only flips bits 5-6 of the chunk field (c = 2 fd + (p >> 1), swizzle = c ^ (row & 7)), and the step
// offset 4096 s never reaches them - so the XOR comes AFTER the add.  Four precomputed addresses (what hipcc makes of
// tr_frag in a loop) cost the dK/dV kernel two scratch reloads with an s_waitcnt vmcnt(0) each per k-step at its
// 80-register budget.  Rows must lie inside the tile (the unclamped steps).
__device__ __forceinline__ uint32_t tr_lane0(const char* tile, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    return (uint32_t)(uintptr_t)tile + (uint32_t)(rm2(4 * g + q, p >> 1) + (p & 1) * 8);
}
template <int FD>
__device__ __forceinline__ bf16x8 tr_frag_x(uint32_t lane_step, int tile_off) {
    typedef __attribute__((address_space(3))) s16x4 lds_v;
    uint32_t a = lane_step ^ (uint32_t)(FD << 5);
    // (the optimiser must not fold the XOR into four loop-invariant addresses again)
    asm volatile("" : "+v"(a));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(a + tile_off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(uintptr_t)(a + tile_off + 2048));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// (b, h) pair and half of a block id: ids b and b + 8 (one XCD under round-robin placement: speed only) are the two
// halves of one pair
// (P parts per pair instead of two halves: ids b, b + 8, ..., b + 8 (P - 1))
template <int P> __device__ __forceinline__ void unit_of_block(int bid, int& bh, int& part) {
    bh = (bid / (8 * P)) * 8 + (bid & 7);
    part = (bid >> 3) % P;
}

template <typename F> __device__ __forceinline__ void static_for4(F&& f) {
    f(std::integral_constant<int, 0>{});
    f(std::integral_constant<int, 1>{});
    f(std::integral_constant<int, 2>{});
    f(std::integral_constant<int, 3>{});
}

// waves per block = 16-row tiles of one part of a head (L <= 224: 14 tiles): P = 2 halves of 7 waves, three blocks per CU at
// <= 80 registers; P = 4 quarters of 4 waves, three blocks per CU = 3 waves per SIMD at <= 168 registers, two full rounds
// of 768 blocks (FFM_ATTN_PARTS=4: measured in round 3, see run_fwd2s)
template <int P> constexpr int a2_nw() { return P == 2 ? 7 : 4; }
template <int P> constexpr int a2_wpe() { return P == 2 ? 6 : 3; }
constexpr float A2_C = 0.125f * 1.44269504088896341f;

template <int NF, bool SH, int P = 2>
__global__ __launch_bounds__(a2_nw<P>() * 64) __attribute__((amdgpu_waves_per_eu(a2_wpe<P>(), a2_wpe<P>()))) void attn2_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                               float* __restrict__ lse, int L, int heads, int BH) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int CA = 8, CB = NF - CA;                           // key fragments of the two chunks
    static_assert(NF >= 9 && NF <= 14, "two chunks: 8 fragments + 1..6");
    // rows of a tile: 16 NF, or 8 fewer when L leaves the last 8 rows of the last fragment empty (L = 197: 200 rows,
    // 51,200 B for both tiles, three blocks per CU); row indices beyond are clamped
    constexpr int R8 = SH ? NF * 16 - 8 : NF * 16, rmax = R8 - 1;
    char* Vs = smem;                                              // [R8][128 B]
    char* Ks = smem + R8 * 128;
    int bh, half;
    constexpr int A2_NW = a2_nw<P>();
    unit_of_block<P>(blockIdx.x, bh, half);
    if (bh >= BH) return;
    const int b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const bf16_t* base = qkv + (size_t)b * L * ld + h * HD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int NFq = (L + 15) >> 4, tb = NFq / P, tr = NFq % P;    // tiles of part p: tb + (p < tr), from p tb + min(p, tr)
    const int qt = half * tb + (half < tr ? half : tr) + wave;    // this wave's 16-query tile
    const bool active = wave < tb + (half < tr ? 1 : 0);

    dma_tile<A2_NW>(base + 2 * E, ld, L, R8, Vs, wave, lane);
    dma_tile<A2_NW>(base + E, ld, L, R8, Ks, wave, lane);
    const int q = qt * 16 + col;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = gfrag<bf16_t>(base, ld, active ? q : 0, L, ks, g);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active || (FFM_ATTN_ABL & 1)) return;                                          // whole waves only (the transposed reads need EXEC = all ones)

    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) o[fd] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](auto F0_, auto CF_) {
        constexpr int F0 = decltype(F0_)::value, CF = decltype(CF_)::value, CP = (CF + 1) & ~1;
        f32x4 s[CP];
        float mx = m;
#pragma unroll
        for (int f = 0; f < CF; ++f) {
            const int row = (F0 + f) * 16 + col;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(Ks + rm2(row, ks * 4 + g)), qf[ks], acc, 0, 0, 0);
            // only the LAST key fragment can straddle L (NF = ceil(L / 16)): a compile-time test in the unrolled loop - as a
            // run-time one hipcc turned it into compare + select on every element of every fragment
            if (F0 + f == NF - 1 && (F0 + f) * 16 + 15 >= L) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((F0 + f) * 16 + g * 4 + e >= L) acc[e] = -INFINITY;
            }
            s[f] = acc;
            // (hipcc puts a canonicalising v_max_f32 x, x in front of fmaxf on every MFMA result; v_max3_f32 by inline asm
            // is NOT the way out: the hazard recogniser does not see an asm statement as a reader of the MFMA's registers,
            // inserts no wait states in front of it, and the maximum comes out as garbage - measured, round 3)
            mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
        }
        if constexpr (CP > CF) s[CF] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        mx = group4_max(mx);
        // running maximum: everything accumulated so far is rescaled by 2^(c (m - mx)) (0 the first time: m = -inf)
        const float alpha = __builtin_amdgcn_exp2f((m - mx) * A2_C);
        const float mc = mx * A2_C;
        m = mx;
        l *= alpha;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[fd][e] *= alpha;
        float sum = 0.f;
#pragma unroll
        for (int st = 0; st < CP / 2; ++st) {
            bf16x8 pf;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[2 * st + hf][e], A2_C, -mc));
                    sum += p;
                    pf[4 * hf + e] = (bf16_t)p;
                }
#pragma unroll
            for (int fd = 0; fd < 4; ++fd)
                o[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<false>(Vs, F0 / 2 + st, fd, lane, rmax), pf, o[fd], 0, 0, 0);
        }
        l += group4_sum(sum);
    };
    chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, CA>{});
    chunk(std::integral_constant<int, CA>{}, std::integral_constant<int, CB>{});

    if (q < L) {
        const float inv = 1.0f / l;
        bf16_t* orow = out + ((size_t)b * L + q) * E + h * HD;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            f32x4 v = o[fd];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= inv;
            Vec4<bf16_t>::store(orow + fd * 16 + g * 4, v);
        }
        if (g == 0 && lse) lse[((size_t)b * heads + h) * L + q] = m * 0.125f + __logf(l);
    }
}

// FFM_ATTN_PARTS=4 (A/B runs): quarter heads of 4 waves instead of half heads of 7
inline int attn_parts() {
    static const int p = (getenv("FFM_ATTN_PARTS") && getenv("FFM_ATTN_PARTS")[0] == '4') ? 4 : 2;
    return p;
}

template <int NF, bool SH, int P>
int run_fwd2p(const void* qkv, void* out, float* lse, int B, int L, int heads, hipStream_t s) {
    constexpr int lds = 2 * (SH ? NF * 16 - 8 : NF * 16) * 128;
    int e = set_lds(attn2_fwd_kernel<NF, SH, P>, lds);
    if (e) return e;
    const int BH = B * heads;
    hipLaunchKernelGGL((attn2_fwd_kernel<NF, SH, P>), dim3(((BH + 7) / 8) * 8 * P), dim3(a2_nw<P>() * 64), lds, s, (const bf16_t*)qkv,
                       (bf16_t*)out, lse, L, heads, BH);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
template <int NF, bool SH>
int run_fwd2s(const void* qkv, void* out, float* lse, int B, int L, int heads, hipStream_t s) {
    return attn_parts() == 4 ? run_fwd2p<NF, SH, 4>(qkv, out, lse, B, L, heads, s) : run_fwd2p<NF, SH, 2>(qkv, out, lse, B, L, heads, s);
}
template <int NF>
int run_fwd2(const void* qkv, void* out, float* lse, int B, int L, int heads, hipStream_t s) {
    return run_fwd2s<NF, false>(qkv, out, lse, B, L, heads, s);
}


constexpr float A2_LOG2E = 1.44269504088896341f;

// dQ (and delta = rowsum(dO * O), which the dK/dV kernel reads back): a wave owns a 16-query tile, the key on the MFMA
// row.  S^T = K Q^T and dP^T = V dO^T share the lane layout; dP's accumulator starts at -delta, so dS^T = p * acc with
// p = exp2(c s - log2(e) lse): one FMA, one exp and one multiply per element (the 1/8 goes onto the dQ tile at the end).
template <int NF, bool SH, int P = 2>
__global__ __launch_bounds__(a2_nw<P>() * 64) __attribute__((amdgpu_waves_per_eu(a2_wpe<P>(), a2_wpe<P>()))) void attn2_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                  const float* __restrict__ lse, const bf16_t* __restrict__ o_fwd,
                                                                  bf16_t* __restrict__ dqkv, float* __restrict__ delta, int L,
                                                                  int heads, int BH) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int R8 = SH ? NF * 16 - 8 : NF * 16, rmax = R8 - 1;
    char* Ks = smem;
    char* Vs = smem + R8 * 128;
    int bh, half;
    constexpr int A2_NW = a2_nw<P>();
    unit_of_block<P>(blockIdx.x, bh, half);
    if (bh >= BH) return;
    const int b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const bf16_t* base = qkv + (size_t)b * L * ld + h * HD;
    const bf16_t* dob = d_o + (size_t)b * L * E + h * HD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int NFq = (L + 15) >> 4, tb = NFq / P, tr = NFq % P;
    const int qt = half * tb + (half < tr ? half : tr) + wave;
    const bool active = wave < tb + (half < tr ? 1 : 0);

    dma_tile<A2_NW>(base + E, ld, L, R8, Ks, wave, lane);
    dma_tile<A2_NW>(base + 2 * E, ld, L, R8, Vs, wave, lane);
    const int q = qt * 16 + col, qs = active ? q : 0;
    bf16x8 qf[2], dof[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = gfrag<bf16_t>(base, ld, qs, L, ks, g);
        dof[ks] = gfrag<bf16_t>(dob, E, qs, L, ks, g);
        of[ks] = gfrag<bf16_t>(o_fwd + (size_t)b * L * E + h * HD, E, qs, L, ks, g);
    }
    const float lq2 = lse[((size_t)b * heads + h) * L + (qs < L ? qs : L - 1)] * A2_LOG2E;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active || (FFM_ATTN_ABL & 1)) return;

    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)of[ks][e] * (float)dof[ks][e];
    dl = group4_sum(dl);
    if (g == 0 && q < L) delta[((size_t)b * heads + h) * L + q] = dl;

    f32x4 dq[4];
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) dq[fd] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NS = (NF + 1) / 2;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        bf16x8 df;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            constexpr int dummy = 0;
            (void)dummy;
            const int f = 2 * st + hf;
            if (f < NF) {
                const int row = f * 16 + col;
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {-dl, -dl, -dl, -dl};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(Ks + rm2(row, ks * 4 + g)), qf[ks], sa, 0, 0, 0);
                    pa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(Vs + rm2(row, ks * 4 + g)), dof[ks], pa, 0, 0, 0);
                }
                if (f == NF - 1 && f * 16 + 15 >= L) {               // (compile-time: only the last fragment can straddle L)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f * 16 + g * 4 + e >= L) sa[e] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) df[4 * hf + e] = (bf16_t)(__builtin_amdgcn_exp2f(fmaf(sa[e], A2_C, -lq2)) * pa[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) df[4 * hf + e] = (bf16_t)0.f;
            }
        }
#pragma unroll
        for (int fd = 0; fd < 4; ++fd)
            dq[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<false>(Ks, st, fd, lane, rmax), df, dq[fd], 0, 0, 0);
    }
    if (q < L) {
        bf16_t* drow = dqkv + ((size_t)b * L + q) * ld + h * HD;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            f32x4 v = dq[fd];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= 0.125f;
            Vec4<bf16_t>::store(drow + fd * 16 + g * 4, v);
        }
    }
}

// dK / dV: a wave owns a 16-key tile and sweeps the queries, the query on the MFMA row: S = Q K^T, dP = dO V^T (its
// accumulator starts at -delta[q]), then dV^T += dO^T P and dK^T += Q^T dS with the transposed operands read from the
// same row-major Q / dO tiles (ds_read_b64_tr_b16).  lse (times log2 e) and delta sit in LDS per query.
template <int NF, bool SH, int P = 2>
__global__ __launch_bounds__(a2_nw<P>() * 64) __attribute__((amdgpu_waves_per_eu(a2_wpe<P>(), a2_wpe<P>()))) void attn2_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, int L, int heads, int BH) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int R8 = SH ? NF * 16 - 8 : NF * 16, rmax = R8 - 1;
    char* Qs = smem;
    char* dOs = smem + R8 * 128;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * R8 * 128);     // [NF * 16]
    float* del_s = lse_s + NF * 16;
    int bh, half;
    constexpr int A2_NW = a2_nw<P>();
    unit_of_block<P>(blockIdx.x, bh, half);
    if (bh >= BH) return;
    const int b = bh / heads, h = bh % heads;
    const int E = heads * HD, ld = 3 * E;
    const bf16_t* base = qkv + (size_t)b * L * ld + h * HD;
    const bf16_t* dob = d_o + (size_t)b * L * E + h * HD;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int NFq = (L + 15) >> 4, tb = NFq / P, tr = NFq % P;
    const int kt = half * tb + (half < tr ? half : tr) + wave;
    const bool active = wave < tb + (half < tr ? 1 : 0);

    dma_tile<A2_NW>(base, ld, L, R8, Qs, wave, lane);
    dma_tile<A2_NW>(dob, E, L, R8, dOs, wave, lane);
    for (int i = tid; i < NF * 16; i += A2_NW * 64) {
        const size_t o = ((size_t)b * heads + h) * L + i;
        lse_s[i] = i < L ? lse[o] * A2_LOG2E : 0.f;
        del_s[i] = i < L ? delta[o] : 0.f;
    }
    const int key = kt * 16 + col, kc = active ? key : 0;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = gfrag<bf16_t>(base + E, ld, kc, L, ks, g);
        vf[ks] = gfrag<bf16_t>(base + 2 * E, ld, kc, L, ks, g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active || (FFM_ATTN_ABL & 1)) return;

    f32x4 dv[4], dk[4];
#pragma unroll
    for (int fd = 0; fd < 4; ++fd) { dv[fd] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[fd] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    constexpr int NS = (NF + 1) / 2;
    const uint32_t tr0 = tr_lane0(Qs, lane);                          // (dOs = Qs + R8 * 128: an immediate offset)
    // one k-step = 32 queries (fragments 2 st, 2 st + 1).  The loop is NOT unrolled (unrolled, the compiler hoists the
    // reads of several steps and the kernel leaves the 80 registers that three blocks per CU allow); the last step is
    // peeled: it alone can touch rows beyond the tile or a phantom fragment.
    auto step = [&](int st, auto LAST_) {
        constexpr bool LAST = decltype(LAST_)::value;
        bf16x8 pf, df;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int f = 2 * st + hf;
            if (!LAST || f < NF) {
                int row = f * 16 + col;
                if constexpr (LAST) row = row < rmax ? row : rmax;
                const int qb = f * 16 + g * 4;                    // this lane's 4 consecutive queries
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&lse_s[qb]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&del_s[qb]);
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {-d4[0], -d4[1], -d4[2], -d4[3]};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(Qs + rm2(row, ks * 4 + g)), kf[ks], sa, 0, 0, 0);
                    pa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(dOs + rm2(row, ks * 4 + g)), vf[ks], pa, 0, 0, 0);
                }
                if constexpr (LAST) {                             // queries beyond L contribute nothing
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (qb + e >= L) sa[e] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sa[e], A2_C, -l4[e]));
                    pf[4 * hf + e] = (bf16_t)p;
                    df[4 * hf + e] = (bf16_t)(p * pa[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { pf[4 * hf + e] = (bf16_t)0.f; df[4 * hf + e] = (bf16_t)0.f; }
            }
        }
        if constexpr (LAST) {
#pragma unroll
            for (int fd = 0; fd < 4; ++fd) {
                dv[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<true>(dOs, st, fd, lane, rmax), pf, dv[fd], 0, 0, 0);
                dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<true>(Qs, st, fd, lane, rmax), df, dk[fd], 0, 0, 0);
            }
        } else {
            const uint32_t ls = tr0 + (uint32_t)st * 4096u;          // one lane register for all eight transposed reads
            static_for4([&](auto FD_) {
                constexpr int fd = decltype(FD_)::value;
                dv[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_x<fd>(ls, R8 * 128), pf, dv[fd], 0, 0, 0);
                dk[fd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag_x<fd>(ls, 0), df, dk[fd], 0, 0, 0);
            });
        }
    };
#pragma unroll 1
    for (int st = 0; st < NS - 1; ++st) step(st, std::false_type{});
    step(NS - 1, std::true_type{});
    if (key < L) {
        bf16_t* drow = dqkv + ((size_t)b * L + key) * ld + h * HD;
#pragma unroll
        for (int fd = 0; fd < 4; ++fd) {
            f32x4 v = dk[fd];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= 0.125f;
            Vec4<bf16_t>::store(drow + E + fd * 16 + g * 4, v);
            Vec4<bf16_t>::store(drow + 2 * E + fd * 16 + g * 4, dv[fd]);
        }
    }
}

template <int NF, bool SH, int P>
int run_bwd2p(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L,
              int heads, hipStream_t s) {
    constexpr int R8 = SH ? NF * 16 - 8 : NF * 16;
    constexpr int lds_dq2 = 2 * NF * 16 * 128, lds_dkv2 = 2 * R8 * 128 + 2 * NF * 16 * 4;
    const int BH = B * heads;
    int e = set_lds(attn2_bwd_dq_kernel<NF, false, P>, lds_dq2);
    if (e) return e;
    e = set_lds(attn2_bwd_dkv_kernel<NF, SH, P>, lds_dkv2);
    if (e) return e;
    const dim3 grid(((BH + 7) / 8) * 8 * P), block(a2_nw<P>() * 64);
    hipLaunchKernelGGL((attn2_bwd_dq_kernel<NF, false, P>), grid, block, lds_dq2, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       (const bf16_t*)out, (bf16_t*)dqkv, delta, L, heads, BH);
    FFM_CHECK_LAUNCH();
    hipLaunchKernelGGL((attn2_bwd_dkv_kernel<NF, SH, P>), grid, block, lds_dkv2, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse,
                       (const float*)delta, (bf16_t*)dqkv, L, heads, BH);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
template <int NF, bool SH>
int run_bwd2s(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L,
              int heads, hipStream_t s) {
    return attn_parts() == 4 ? run_bwd2p<NF, SH, 4>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s)
                             : run_bwd2p<NF, SH, 2>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
}
template <int NF>
int run_bwd2(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L,
             int heads, hipStream_t s) {
    return L <= NF * 16 - 8 ? run_bwd2s<NF, true>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s)
                            : run_bwd2s<NF, false>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
}

// FFM_ATTN=v1: always the first-generation kernels; FFM_ATTN=v2: the second generation where it applies (A/B runs).
// Default: the third generation (attention3.hip: fat waves on 32x32x16 MFMAs) for 65..256 unmasked tokens in 16-bit storage.
inline int attn_gen_forced() {
    static const int g = (getenv("FFM_ATTN") && getenv("FFM_ATTN")[0] == 'v' && getenv("FFM_ATTN")[1] >= '1' && getenv("FFM_ATTN")[1] <= '3')
                             ? getenv("FFM_ATTN")[1] - '0' : 0;
    return g;
}
inline bool attn_v1_forced() { return attn_gen_forced() == 1; }
inline bool attn3_allowed() { return attn_gen_forced() == 0 || attn_gen_forced() == 3; }

inline int pick_split(int bh, int NF) {
    // aim for >= ~3 blocks per CU while keeping >= 4 tiles (one per wave) per block
    int s = 1;
    while (bh * s < 768 && (NF + s) / (s + 1) >= 4) ++s;
    return s;
}

template <typename T, int NFP>
int run_fwd(const void* qkv, void* out, float* lse, int B, int L, int heads, int causal, hipStream_t s) {
    const int NF = (L + 15) / 16;
    const int lds = lds_fwd<T>(NFP);
    if (sizeof(T) == 2 && B * heads >= 256 && NF >= 8) {
        // enough (b, h) pairs to fill the chip: one wave per 16-query tile and no q-split, so K / V are staged once
        // per pair and no wave runs two tiles back to back
        constexpr int NTH = NFP <= 14 ? 64 * NFP : 512;
        int e = set_lds(attn_fwd_kernel<T, NFP, NTH>, lds);
        if (e) return e;
        hipLaunchKernelGGL((attn_fwd_kernel<T, NFP, NTH>), dim3(B * heads, 1), dim3(NTH), lds, s, (const T*)qkv, (T*)out,
                           lse, L, heads, causal, 1);
        FFM_CHECK_LAUNCH();
        return FFM_OK;
    }
    const int split = pick_split(B * heads, NF);
    int e = set_lds(attn_fwd_kernel<T, NFP, 256>, lds);
    if (e) return e;
    hipLaunchKernelGGL((attn_fwd_kernel<T, NFP, 256>), dim3(B * heads, split), dim3(256), lds, s, (const T*)qkv, (T*)out,
                       lse, L, heads, causal, split);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

template <typename T, int NFP>
int run_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
            int L, int heads, int causal, hipStream_t s) {
    (void)delta;          // kept in the ABI as scratch; the row sums of dO * O are formed inside the two kernels
    int lds = lds_dq<T>(NFP);
    int e = set_lds(attn_bwd_dq_kernel<T, NFP, kStaged<T>(), kBwThreads<T, NFP>()>, lds);
    if (e) return e;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, NFP, kStaged<T>(), kBwThreads<T, NFP>()>), dim3(B * heads), dim3(kBwThreads<T, NFP>()), lds, s,
                       (const T*)qkv, (const T*)dout, lse, (const T*)out, (T*)dqkv, L, heads, causal);
    FFM_CHECK_LAUNCH();
    lds = lds_dkv<T>(NFP);
    e = set_lds(attn_bwd_dkv_kernel<T, NFP, kStaged<T>(), kBwThreads<T, NFP>()>, lds);
    if (e) return e;
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, NFP, kStaged<T>(), kBwThreads<T, NFP>()>), dim3(B * heads), dim3(kBwThreads<T, NFP>()), lds, s,
                       (const T*)qkv, (const T*)dout, lse, (const T*)out, (T*)dqkv, L, heads, causal);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}

#define NFP_SWITCH(CALL)                 \
    switch (nfp) {                       \
        case 2: return CALL(2);          \
        case 4: return CALL(4);          \
        case 6: return CALL(6);          \
        case 8: return CALL(8);          \
        case 10: return CALL(10);        \
        case 12: return CALL(12);        \
        case 14: return CALL(14);        \
        case 16: return CALL(16);        \
        default: return FFM_EUNSUP;      \
    }

template <typename T>
int dispatch_fwd(int nfp, const void* qkv, void* out, float* lse, int B, int L, int heads, int causal, hipStream_t s) {
#define CALL(N) run_fwd<T, N>(qkv, out, lse, B, L, heads, causal, s)
    NFP_SWITCH(CALL)
#undef CALL
}
template <typename T>
int dispatch_bwd(int nfp, const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                 void* dqkv, int B, int L, int heads, int causal, hipStream_t s) {
#define CALL(N) run_bwd<T, N>(qkv, out, dout, lse, delta, dqkv, B, L, heads, causal, s)
    NFP_SWITCH(CALL)
#undef CALL
}

}  // namespace

// attention3.hip
int ffm_attn3_fwd(const void* qkv, void* out, float* lse, int B, int L, int heads, int dtype, hipStream_t s);
int ffm_attn3_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B, int L, int heads,
                  int dtype, hipStream_t s, const float* ln_wg, const float* ln_d, float* ln_part);

extern "C" int ffm_attention_fwd(const void* qkv, void* out, float* lse, int B, int L, int heads, int causal,
                                 int dtype, void* stream) {
    if (!qkv || !out || B <= 0 || L <= 0 || heads <= 0) return FFM_EINVAL;
    if (L > 256) return FFM_EUNSUP;
    if (((uintptr_t)qkv | (uintptr_t)out) & 15) return FFM_EINVAL;
    const int nfp = (((L + 15) / 16) + 1) & ~1;
    hipStream_t s = (hipStream_t)stream;
    if ((dtype == FFM_BF16 || dtype == FFM_F16) && !causal && attn3_allowed()) {
        const int e = ffm_attn3_fwd(qkv, out, lse, B, L, heads, dtype, s);
        if (e != FFM_EUNSUP) return e;
    }
    if (dtype == FFM_BF16 && !causal && !attn_v1_forced()) {
        switch ((L + 15) / 16) {                        // the vision tower's shapes: attn2_* (half a head per block)
            case 9: return run_fwd2<9>(qkv, out, lse, B, L, heads, s);
            case 10: return run_fwd2<10>(qkv, out, lse, B, L, heads, s);
            case 11: return run_fwd2<11>(qkv, out, lse, B, L, heads, s);
            case 12: return run_fwd2<12>(qkv, out, lse, B, L, heads, s);
            case 13: return run_fwd2<13>(qkv, out, lse, B, L, heads, s);
            case 14: return run_fwd2<14>(qkv, out, lse, B, L, heads, s);
        }
    }
    if (dtype == FFM_BF16) return dispatch_fwd<bf16_t>(nfp, qkv, out, lse, B, L, heads, causal, s);
    if (dtype == FFM_F32) return dispatch_fwd<float>(nfp, qkv, out, lse, B, L, heads, causal, s);
    return FFM_EINVAL;
}

extern "C" int ffm_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                 void* dqkv, int B, int L, int heads, int causal, int dtype, void* stream) {
    if (!qkv || !out || !dout || !lse || !delta || !dqkv || B <= 0 || L <= 0 || heads <= 0) return FFM_EINVAL;
    if (L > 256) return FFM_EUNSUP;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) return FFM_EINVAL;
    const int nfp = (((L + 15) / 16) + 1) & ~1;
    hipStream_t s = (hipStream_t)stream;
    if ((dtype == FFM_BF16 || dtype == FFM_F16) && !causal && attn3_allowed()) {
        const int e = ffm_attn3_bwd(qkv, out, dout, lse, delta, dqkv, B, L, heads, dtype, s, nullptr, nullptr, nullptr);
        if (e != FFM_EUNSUP) return e;
    }
    if (dtype == FFM_BF16 && !causal && !attn_v1_forced()) {
        switch ((L + 15) / 16) {
            case 9: return run_bwd2<9>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
            case 10: return run_bwd2<10>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
            case 11: return run_bwd2<11>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
            case 12: return run_bwd2<12>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
            case 13: return run_bwd2<13>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
            case 14: return run_bwd2<14>(qkv, out, dout, lse, delta, dqkv, B, L, heads, s);
        }
    }
    if (dtype == FFM_BF16) return dispatch_bwd<bf16_t>(nfp, qkv, out, dout, lse, delta, dqkv, B, L, heads, causal, s);
    if (dtype == FFM_F32) return dispatch_bwd<float>(nfp, qkv, out, dout, lse, delta, dqkv, B, L, heads, causal, s);
    return FFM_EINVAL;
}

// ffm_attention_bwd that ALSO leaves the two row sums of the LayerNorm backward behind the in-projection's dX product
// (include/ffm_hip.h): ln_part [2 heads][B L][2] - slot h: the q columns of head h (dQ kernel), slot heads + h: its k and v
// columns (dK/dV kernel).  Served by the third-generation kernels only (16-bit storage, no mask, 65..256 tokens):
// FFM_EUNSUP otherwise, and nothing is launched
// (ffm_attention_bwd_lnstat_ok says so beforehand).
extern "C" int ffm_attention_bwd_lnstat_ok(int L, int causal, int dtype) {
    // (97..256 tokens: the dK/dV kernel lays its four 64-float tables over the row constants of at least four 32-token tiles)
    return ((dtype == FFM_BF16 || dtype == FFM_F16) && !causal && attn3_allowed() && L > 96 && L <= 256) ? 1 : 0;
}

extern "C" int ffm_attention_bwd_lnstat(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                        const float* ln_wg, const float* ln_d, float* ln_part, int B, int L, int heads, int causal,
                                        int dtype, void* stream) {
    if (!qkv || !out || !dout || !lse || !delta || !dqkv || !ln_wg || !ln_d || !ln_part || B <= 0 || L <= 0 || heads <= 0) return FFM_EINVAL;
    if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv | (uintptr_t)ln_wg | (uintptr_t)ln_d) & 15) return FFM_EINVAL;
    if (!ffm_attention_bwd_lnstat_ok(L, causal, dtype)) return FFM_EUNSUP;
    return ffm_attn3_bwd(qkv, out, dout, lse, delta, dqkv, B, L, heads, dtype, (hipStream_t)stream, ln_wg, ln_d, ln_part);
}
