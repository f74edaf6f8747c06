// On-device evaluator counts: everything the reference's binary-task evaluator reports
// (evaluation/evaluator_oph.py:37-150, evaluation/metrics.py:197-311, 513-552) is a function of a few INTEGER counts
// per demographic group, so one pass over the test scores on the GPU replaces the sort-based host metrics:
//   AUC (sklearn roc_auc_score == Mann-Whitney with ties at 1/2) = (wins + ties/2) / (n_pos * n_neg)
//   accuracy / macro-F1 / DPD / EOD                              = ratios of TP, FP, TN, FN
// Exact integer arithmetic: the result does not depend on launch geometry or summation order.
#include "common.h"

namespace {

constexpr int EV_NPOS = 0, EV_NNEG = 1, EV_WIN1 = 2, EV_TIE1 = 3, EV_WIN0 = 4, EV_TIE0 = 5, EV_TP = 6, EV_FP = 7, EV_TN = 8,
              EV_FN = 9;
constexpr int EV_SLOTS = 10;                 // FFM_EVAL_SLOTS in the header
constexpr int EV_MAXG = FFM_MAX_GROUPS;      // groups 0..G-1, slot G = attribute -1 (unknown), slot G+1 = all samples

typedef unsigned long long u64;

// thread = one sample i; it walks the j range of its grid row through LDS tiles and counts, for a POSITIVE i against
// every NEGATIVE j, the class-1 column comparison p1_i vs p1_j and the class-0 column comparison p0_j vs p0_i
// (class 0's positives are the label-0 samples): the two one-vs-rest AUCs the reference averages.
__global__ __launch_bounds__(256) void eval_pairs_kernel(const float* __restrict__ prob, const int64_t* __restrict__ label,
                                                         const int64_t* __restrict__ attr, int N, int G, int jchunk,
                                                         u64* __restrict__ out) {
    __shared__ float sp0[256], sp1[256];
    __shared__ int sg[256];                  // group slot of j, or -1 for a positive j (skipped)
    __shared__ u64 acc[(EV_MAXG + 2) * EV_SLOTS];
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    for (int k = tid; k < (G + 2) * EV_SLOTS; k += 256) acc[k] = 0;
    float p0 = 0.f, p1 = 0.f;
    int gi = -1;
    bool pos = false;
    if (i < N) {
        p0 = prob[2 * (size_t)i];
        p1 = prob[2 * (size_t)i + 1];
        pos = label[i] == 1;
        const int64_t a = attr ? attr[i] : -1;
        gi = (a >= 0 && a < G) ? (int)a : G;
    }
    __syncthreads();
    if (blockIdx.y == 0 && i < N) {                      // per-sample counts, once
        const bool pred1 = p1 > p0;                      // argmax with the first index winning ties
        const int what = pos ? (pred1 ? EV_TP : EV_FN) : (pred1 ? EV_FP : EV_TN);
        atomicAdd(&acc[gi * EV_SLOTS + (pos ? EV_NPOS : EV_NNEG)], 1ull);
        atomicAdd(&acc[gi * EV_SLOTS + what], 1ull);
        atomicAdd(&acc[(G + 1) * EV_SLOTS + (pos ? EV_NPOS : EV_NNEG)], 1ull);
        atomicAdd(&acc[(G + 1) * EV_SLOTS + what], 1ull);
    }
    unsigned w1a = 0, t1a = 0, w0a = 0, t0a = 0, w1g = 0, t1g = 0, w0g = 0, t0g = 0;
    const int j0 = blockIdx.y * jchunk, j1 = (j0 + jchunk) < N ? (j0 + jchunk) : N;
    for (int jt = j0; jt < j1; jt += 256) {
        const int j = jt + tid;
        __syncthreads();
        if (j < j1) {
            sp0[tid] = prob[2 * (size_t)j];
            sp1[tid] = prob[2 * (size_t)j + 1];
            const int64_t a = attr ? attr[j] : -1;
            sg[tid] = label[j] == 1 ? -1 : ((a >= 0 && a < G) ? (int)a : G);
        } else {
            sg[tid] = -1;
        }
        __syncthreads();
        if (pos) {
#pragma unroll 8
            for (int k = 0; k < 256; ++k) {
                const int gj = sg[k];
                const bool neg = gj >= 0, same = gj == gi;
                const float q0 = sp0[k], q1 = sp1[k];
                const unsigned a1 = neg && (p1 > q1), b1 = neg && (p1 == q1), a0 = neg && (q0 > p0), b0 = neg && (q0 == p0);
                w1a += a1; t1a += b1; w0a += a0; t0a += b0;
                w1g += same && a1; t1g += same && b1; w0g += same && a0; t0g += same && b0;
            }
        }
    }
    if (pos) {
        u64* all = &acc[(G + 1) * EV_SLOTS];
        u64* grp = &acc[gi * EV_SLOTS];
        if (w1a) atomicAdd(&all[EV_WIN1], (u64)w1a);
        if (t1a) atomicAdd(&all[EV_TIE1], (u64)t1a);
        if (w0a) atomicAdd(&all[EV_WIN0], (u64)w0a);
        if (t0a) atomicAdd(&all[EV_TIE0], (u64)t0a);
        if (w1g) atomicAdd(&grp[EV_WIN1], (u64)w1g);
        if (t1g) atomicAdd(&grp[EV_TIE1], (u64)t1g);
        if (w0g) atomicAdd(&grp[EV_WIN0], (u64)w0g);
        if (t0g) atomicAdd(&grp[EV_TIE0], (u64)t0g);
    }
    __syncthreads();
    for (int k = tid; k < (G + 2) * EV_SLOTS; k += 256)
        if (acc[k]) atomicAdd(&out[k], acc[k]);
}

}  // namespace

extern "C" int ffm_eval_counts(const float* prob, const int64_t* label, const int64_t* attr, int N, int G, uint64_t* out,
                               void* stream) {
    if (!prob || !label || !out || N <= 0 || G < 0 || G > EV_MAXG) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const hipError_t me = hipMemsetAsync(out, 0, sizeof(uint64_t) * (size_t)(G + 2) * EV_SLOTS, s);
    if (me != hipSuccess) return (int)me;
    const int bi = (N + 255) / 256;
    // enough blocks to fill the chip: split the j range until there are ~1024 blocks; every thread keeps 32-bit
    // counters, so a chunk never exceeds 2^31 comparisons
    int ysplit = (1024 + bi - 1) / bi;
    const int maxsplit = (N + 255) / 256;
    if (ysplit > maxsplit) ysplit = maxsplit;
    if (ysplit < 1) ysplit = 1;
    int jchunk = ((N + ysplit - 1) / ysplit + 255) / 256 * 256;
    ysplit = (N + jchunk - 1) / jchunk;
    hipLaunchKernelGGL(eval_pairs_kernel, dim3(bi, ysplit), dim3(256), 0, s, prob, label, attr, N, G, jchunk,
                       (u64*)out);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
