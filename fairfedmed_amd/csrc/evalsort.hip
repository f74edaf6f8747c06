// Evaluator counts for LARGE test sets (ffm_eval_counts_sorted): the same integer table as ffm_eval_counts
// (evalmetrics.hip: one thread per positive sample against every negative one, N^2 / 2 compares - fine at 20 k
// samples, 100 x slower than a sort at 200 k), in O(N log N):
//   per probability column c and per scope (all samples / the sample's own group):
//     64-bit key = scope group << 32 | order-preserving image of the float score, radix-sorted with the sample index;
//     E = exclusive prefix count of that column's NEGATIVES in sorted order;
//     a positive with key k wins against E[lower_bound(k)] negatives and ties with E[upper_bound(k)] - E[lower_bound(k)].
// Exact integers, independent of launch geometry, bit-identical to the pair kernel (tests/test_evaluator_gpu.py).
// The sort and the scan are rocPRIM's (device-wide primitives, like the vendor GEMM for the 4-row text glue); keys,
// counting and the table are this file's.  The caller provides the workspace (the library allocates nothing).
#include "common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

constexpr int EV_NPOS = 0, EV_NNEG = 1, EV_WIN1 = 2, EV_TIE1 = 3, EV_WIN0 = 4, EV_TIE0 = 5, EV_TP = 6, EV_FP = 7, EV_TN = 8,
              EV_FN = 9;
constexpr int EV_SLOTS = FFM_EVAL_SLOTS;
typedef unsigned long long u64;

__device__ __forceinline__ uint32_t order_key(float f) {          // a < b  <=>  key(a) < key(b)  (no NaNs, no -0)
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ int group_slot(const int64_t* attr, int i, int G) {
    const int64_t a = attr ? attr[i] : -1;
    return (a >= 0 && a < G) ? (int)a : G;
}

// per-sample counts (n_pos, n_neg, confusion matrix) of the group row and of the 'all' row
__global__ __launch_bounds__(256) void eval_basic_kernel(const float* __restrict__ prob, const int64_t* __restrict__ label,
                                                         const int64_t* __restrict__ attr, int N, int G, u64* __restrict__ out) {
    __shared__ u64 acc[(FFM_MAX_GROUPS + 2) * EV_SLOTS];
    const int tid = threadIdx.x;
    for (int k = tid; k < (G + 2) * EV_SLOTS; k += 256) acc[k] = 0;
    __syncthreads();
    for (int i = blockIdx.x * 256 + tid; i < N; i += gridDim.x * 256) {
        const float p0 = prob[2 * (size_t)i], p1 = prob[2 * (size_t)i + 1];
        const bool pos = label[i] == 1, pred1 = p1 > p0;
        const int gi = group_slot(attr, i, G);
        const int what = pos ? (pred1 ? EV_TP : EV_FN) : (pred1 ? EV_FP : EV_TN);
        atomicAdd(&acc[gi * EV_SLOTS + (pos ? EV_NPOS : EV_NNEG)], 1ull);
        atomicAdd(&acc[gi * EV_SLOTS + what], 1ull);
        atomicAdd(&acc[(G + 1) * EV_SLOTS + (pos ? EV_NPOS : EV_NNEG)], 1ull);
        atomicAdd(&acc[(G + 1) * EV_SLOTS + what], 1ull);
    }
    __syncthreads();
    for (int k = tid; k < (G + 2) * EV_SLOTS; k += 256)
        if (acc[k]) atomicAdd(&out[k], acc[k]);
}

// keys of one (column, scope): column c's score, scope group in the high word (0 for the 'all' scope)
__global__ __launch_bounds__(256) void eval_keys_kernel(const float* __restrict__ prob, const int64_t* __restrict__ attr, int N,
                                                        int G, int c, int grouped, u64* __restrict__ keys,
                                                        uint32_t* __restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const u64 hi = grouped ? (u64)group_slot(attr, i, G) : 0ull;
    keys[i] = (hi << 32) | order_key(prob[2 * (size_t)i + c]);
    idx[i] = (uint32_t)i;
}

// sorted order: 1 for a NEGATIVE of column c (label != c), else 0
__global__ __launch_bounds__(256) void eval_negflag_kernel(const uint32_t* __restrict__ sidx, const int64_t* __restrict__ label,
                                                           int N, int c, uint32_t* __restrict__ flag) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < N) flag[s] = label[sidx[s]] != c ? 1u : 0u;
}

// every POSITIVE of column c (label == c): wins = E[lower_bound(key)], ties = E[upper_bound(key)] - wins.  E has N + 1
// entries (E[N] = all negatives).  The 'all' scope adds to row G + 1, the group scope to the sample's own row.
__global__ __launch_bounds__(256) void eval_count_kernel(const u64* __restrict__ skeys, const uint32_t* __restrict__ sidx,
                                                         const uint32_t* __restrict__ E, const int64_t* __restrict__ label,
                                                         int N, int G, int c, int grouped, u64* __restrict__ out) {
    __shared__ u64 acc[(FFM_MAX_GROUPS + 2) * 2];
    const int tid = threadIdx.x;
    for (int k = tid; k < (G + 2) * 2; k += 256) acc[k] = 0;
    __syncthreads();
    const int s = blockIdx.x * 256 + tid;
    if (s < N && label[sidx[s]] == c) {
        const u64 k = skeys[s];
        int lo = 0, hi = s;                                        // lower bound lies in [0, s]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skeys[mid] < k) lo = mid + 1; else hi = mid;
        }
        const int lb = lo;
        lo = s + 1;                                                // upper bound lies in (s, N]
        hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skeys[mid] <= k) lo = mid + 1; else hi = mid;
        }
        const int ub = lo;
        // (grouped scope: keys of other groups differ in the high word, so both bounds stay inside the group's
        // segment, and E[lb] counts the negatives of EARLIER groups too: subtract the count at the segment start)
        uint32_t base = 0;
        if (grouped) {
            const u64 seg = k & 0xFFFFFFFF00000000ull;
            int a = 0, b = lb;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (skeys[mid] < seg) a = mid + 1; else b = mid;
            }
            base = E[a];
        }
        const uint32_t wins = E[lb] - base, ties = E[ub] - E[lb];
        const int row = grouped ? (int)(k >> 32) : G + 1;
        if (wins) atomicAdd(&acc[row * 2], (u64)wins);
        if (ties) atomicAdd(&acc[row * 2 + 1], (u64)ties);
    }
    __syncthreads();
    const int wslot = c == 1 ? EV_WIN1 : EV_WIN0;
    for (int k = tid; k < (G + 2) * 2; k += 256)
        if (acc[k]) atomicAdd(&out[(k >> 1) * EV_SLOTS + wslot + (k & 1)], acc[k]);
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Ws {
    u64 *k0, *k1;
    uint32_t *i0, *i1, *flag, *E;
    void* temp;
    size_t temp_bytes, total;
};

hipError_t plan(int N, char* base, Ws& w) {
    size_t sort_b = 0, scan_b = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_b, (u64*)nullptr, (u64*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                             (size_t)N, 0u, 36u, (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, scan_b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)N + 1, rocprim::plus<uint32_t>(),
                                (hipStream_t)0);
    if (e != hipSuccess) return e;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
    w.k0 = (u64*)take(sizeof(u64) * N);
    w.k1 = (u64*)take(sizeof(u64) * N);
    w.i0 = (uint32_t*)take(4 * (size_t)N);
    w.i1 = (uint32_t*)take(4 * (size_t)N);
    w.flag = (uint32_t*)take(4 * ((size_t)N + 1));
    w.E = (uint32_t*)take(4 * ((size_t)N + 1));
    w.temp_bytes = sort_b > scan_b ? sort_b : scan_b;
    w.temp = take(w.temp_bytes);
    w.total = off;
    return hipSuccess;
}

}  // namespace

extern "C" int64_t ffm_eval_counts_ws_bytes(int N) {
    if (N <= 0) return FFM_EINVAL;
    Ws w;
    if (plan(N, nullptr, w) != hipSuccess) return FFM_EINVAL;
    return (int64_t)w.total;
}

extern "C" int ffm_eval_counts_sorted(const float* prob, const int64_t* label, const int64_t* attr, int N, int G, uint64_t* out,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!prob || !label || !out || !workspace || N <= 0 || G < 0 || G > FFM_MAX_GROUPS) return FFM_EINVAL;
    if ((uintptr_t)workspace & 255) return FFM_EINVAL;
    Ws w;
    hipError_t e = plan(N, (char*)workspace, w);
    if (e != hipSuccess) return (int)e;
    if ((int64_t)w.total > workspace_bytes) return FFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    e = hipMemsetAsync(out, 0, sizeof(uint64_t) * (size_t)(G + 2) * EV_SLOTS, s);
    if (e != hipSuccess) return (int)e;
    const int nb = (N + 255) / 256;
    hipLaunchKernelGGL(eval_basic_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, prob, label, attr, N, G, (u64*)out);
    FFM_CHECK_LAUNCH();
    for (int c = 1; c >= 0; --c) {
        for (int grouped = 0; grouped < 2; ++grouped) {
            hipLaunchKernelGGL(eval_keys_kernel, dim3(nb), dim3(256), 0, s, prob, attr, N, G, c, grouped, w.k0, w.i0);
            FFM_CHECK_LAUNCH();
            size_t tb = w.temp_bytes;
            e = rocprim::radix_sort_pairs(w.temp, tb, w.k0, w.k1, w.i0, w.i1, (size_t)N, 0u, grouped ? 36u : 32u, s);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(eval_negflag_kernel, dim3(nb), dim3(256), 0, s, w.i1, label, N, c, w.flag);
            FFM_CHECK_LAUNCH();
            e = hipMemsetAsync(w.flag + N, 0, 4, s);
            if (e != hipSuccess) return (int)e;
            tb = w.temp_bytes;
            e = rocprim::exclusive_scan(w.temp, tb, w.flag, w.E, 0u, (size_t)N + 1, rocprim::plus<uint32_t>(), s);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(eval_count_kernel, dim3(nb), dim3(256), 0, s, w.k1, w.i1, w.E, label, N, G, c, grouped, (u64*)out);
            FFM_CHECK_LAUNCH();
        }
    }
    return FFM_OK;
}
