// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ffm_hip.h"

typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
#ifndef FFM_TWIN_F16
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#define FFM_MFMA16_MNEMONIC "v_mfma_f32_16x16x32_bf16"
#else
// The IEEE-half twin of a translation unit (fairfedmed_amd/build.py compiles every 16-bit kernel file a second time with
// -DFFM_TWIN_F16 and links it under renamed symbols): "the 16-bit storage type of this file" becomes _Float16 - the kernels are
// written against bf16_t / bf16x8 as THE 16-bit type, convert by casts only (no bit tricks on the encoding) and name the matrix
// instruction through the two macros below.  The two dtype codes swap so that `dtype == FFM_BF16` selects the 16-bit kernels
// for half inputs; a dtype VALUE passed on to another file is always the caller's real code.
typedef _Float16 bf16_t;
typedef f16x8 bf16x8;
typedef f16x4 bf16x4;
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define FFM_MFMA16_MNEMONIC "v_mfma_f32_16x16x32_f16"
#undef FFM_BF16
#undef FFM_F16
#define FFM_BF16 3
#define FFM_F16 1
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FFM_WAVE 64

#define FFM_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// ---------------------------------------------------------------------------
// Element traits: T is float or bf16_t.  A "chunk" is 16 bytes of a row.
// ---------------------------------------------------------------------------
template <typename T> struct Elem;

template <> struct Elem<float> {
    static constexpr int kPerChunk = 4;      // elements per 16 B
    typedef f32x4 chunk_t;
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};

template <> struct Elem<bf16_t> {
    static constexpr int kPerChunk = 8;
    typedef bf16x8 chunk_t;
    static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
    static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

// 4 consecutive elements <-> 4 floats (8 B for bf16, 16 B for f32)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
        f32x4 v = {(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        return v;
    }
    static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
        bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(p) = r;
    }
};

// 8 consecutive elements <-> 8 floats
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        f32x4 a = {v[0], v[1], v[2], v[3]};
        f32x4 b = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        bf16x8 r = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
        *reinterpret_cast<bf16x8*>(p) = r;
    }
};

// ---------------------------------------------------------------------------
// MFMA 16x16 wrapper.  One "fragment" is 16 bytes per lane of an operand row:
//   bf16: 8 consecutive k of row (lane&15), k-group (lane>>4)      -> 1 MFMA, K = 32
//   f32 : 4 consecutive k of row (lane&15), k-group (lane>>4)      -> 4 MFMAs, K = 16
// For f32 MFMA number e consumes element e of every lane's fragment; which
// physical k that is does not matter as long as both operands use the same
// map (the instruction sums over all k).
// D layout (both): lane holds D[row = 4*(lane>>4) + reg][col = lane&15].
// ---------------------------------------------------------------------------
template <typename T> struct Mma16;

template <> struct Mma16<bf16_t> {
    typedef bf16x8 frag_t;
    static constexpr int kK = 32;  // reduction extent of one fragment pair
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};

template <> struct Mma16<float> {
    typedef f32x4 frag_t;
    static constexpr int kK = 16;
    static __device__ __forceinline__ void mma(f32x4& acc, const frag_t& a, const frag_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------
// wave reductions
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// e^x on the raw v_exp_f32 (1 ulp): __expf / expf add range checks and denormal scaling around the same instruction,
// several VALU ops per element of a softmax tile
__device__ __forceinline__ float fast_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

__device__ __forceinline__ float quick_gelu_f(float x) {
    // clip/model.py:313-315: x * sigmoid(1.702 x)
    return x / (1.0f + expf(-1.702f * x));
}
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
    float s = 1.0f / (1.0f + expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// QuickGELU in the epilogues: exact-ish libm path for the f32 parity mode, hardware exp/rcp for bf16
// (the result is rounded to 8 mantissa bits anyway).
template <typename T> struct Act;
template <> struct Act<float> {
    static __device__ __forceinline__ float gelu(float x) { return quick_gelu_f(x); }
    static __device__ __forceinline__ float gelu_grad(float x) { return quick_gelu_grad_f(x); }
    // both from one call (ffm_gemm_args.gelu_deriv): the same two expressions as above, so fp32 results do not move
    static __device__ __forceinline__ void gelu_both(float x, float& g, float& d) { g = quick_gelu_f(x); d = quick_gelu_grad_f(x); }
};
template <> struct Act<bf16_t> {
    // raw v_exp_f32 / v_rcp_f32 (1 ulp): __expf / __frcp_rn expand to range checks and a full IEEE division
    // (v_div_scale, v_div_fmas, v_div_fixup ...), ~20 instructions per element of a 19-million-element epilogue
    static __device__ __forceinline__ float sigmoid1702(float x) {
        return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.44269504088896341f * x));
    }
    static __device__ __forceinline__ float gelu(float x) { return x * sigmoid1702(x); }
    static __device__ __forceinline__ float gelu_grad(float x) {
        const float s = sigmoid1702(x);
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    static __device__ __forceinline__ void gelu_both(float x, float& g, float& d) {      // one sigmoid for both
        const float s = sigmoid1702(x);
        g = x * s;
        d = s * (1.0f + 1.702f * x * (1.0f - s));
    }
};

// pi_b[g]: 0.7 on the sample's own group, 0.3/(G-1) elsewhere; uniform when
// attr is NULL (trainers/GLP_OT_SVLoRA.py:453-462).
__device__ __forceinline__ float group_mix_w(const int32_t* attr, int sample, int g, int G, float lambda_group) {
    if (attr == nullptr) return 1.0f / (float)G;
    int a = attr[sample];
    return (a == g) ? lambda_group : (1.0f - lambda_group) / (float)(G - 1);
}

// XCD-aware bijective remap of a linear block id (8 XCDs, round-robin
// dispatch): blocks that share an operand panel get consecutive logical ids
// and land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
