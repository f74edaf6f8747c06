// Panel GEMM: kernel selection, the plain / bias / residual instantiations and the load-time weight packer.
// (kernel: gemm_panel_impl.h; FairLoRA epilogues: gemm_panel_rk.hip)
#include "gemm_panel_impl.h"
#include <cstdlib>

// Default set of the newer configurations (tools/bench_panel.py, isolated launches at bs 32; FFM_PANEL_MASK=<int> for A/B):
//   7  208x384 FairLoRA, two waves per SIMD: c_fc forward 53.8 -> 44.7 us, dX(c_proj) 50.5 -> 44.5 us
//   8  160x128 FairLoRA (240 blocks where the 176-row tile launches 216): c_proj forward 40.2 -> 38.1, dX(c_fc) 39.6 -> 38.8
//   10 240x256 plain, two waves per SIMD (243 blocks): qkv forward 31.9 -> 27.7 us (one wave per SIMD, 9: 32.9)
// Measured and no longer instantiated: the 128-column two-wave twins as eight column slabs (5, 6: no gain, twice the LDS
// fragment reads) and the one-wave 240x256 tile (9).  Off but instantiated: the K split
// of the 160x128 tiles (11, 12: 4 column slabs x 2 K halves - isolated c_proj forward 39.6 -> 36.9 us, dX(c_fc) 38.9 ->
// 36.1, dX(qkv) 25.0 -> 23.7, but IN THE STEP, beside the text tower and the LoRA-gradient reductions, the same launches
// take what the 4-wave tiles take (44.0 / 43.8, 40.5 / 38.4 us) and the step is 0.03 ms slower: 4.73 -> 4.77 ms twice in
// one call); tests/test_kernels_gpu.py runs the panel tests with it switched on and on the round-2 tiles alone.
#ifndef FFM_PANEL_MASK_DEFAULT
#define FFM_PANEL_MASK_DEFAULT ((1 << 7) | (1 << 8) | (1 << 10))
#endif

namespace {

// dst[((n16 * K/32 + k32) * 64 + lane) * 8 + i] = src[(n16*16 + (lane & 15)) * ld + k32*32 + (lane >> 4)*8 + i]
__global__ __launch_bounds__(256) void pack_b_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K,
                                                     int ld) {
    const size_t total = (size_t)(N >> 4) * (K >> 5) * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t frag = i >> 6;
        const int k32 = (int)(frag % (size_t)(K >> 5)), n16 = (int)(frag / (size_t)(K >> 5));
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (size_t)(n16 * 16 + (lane & 15)) * ld + k32 * 32 + (lane >> 4) * 8);
        *reinterpret_cast<bf16x8*>(dst + i * 8) = v;
    }
}

// plain epilogues the panel kernel is instantiated for (PANEL_CASE below)
bool plain_flags_ok(int flags) {
    switch (flags) {
        case 0:
        case FFM_EPI_BIAS:
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL:
        case FFM_EPI_BIAS | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS:      // out-proj forward leaving row sums for ln_2
        case FFM_EPI_BIAS | FFM_EPI_LNIN:                              // qkv forward with ln_1 folded in
        case FFM_EPI_LNB_APPLY: return true;                           // dX of qkv applying ln_1's backward
    }
    return false;
}

bool rk_flags_ok(int flags, int rank) {
    if (rank <= 0 || rank > 16) return false;
    if (flags & FFM_EPI_LNB_APPLY)                             // LayerNorm backward applied: the dX epilogue of c_fc only
        return (flags & ~FFM_EPI_RANKOP) == (FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_LNB_APPLY) && rank <= 14;   // (rows 14 / 15 of rk: W gamma, d)
    if (flags & FFM_EPI_LGRAD) {                               // the gradient partial products: the dX epilogue of c_proj only
        // (... which may also leave LayerNorm-backward row sums: FFM_EPI_LNB_STAT rides on the LGRAD epilogue)
        if ((flags & ~(FFM_EPI_RANKOP | FFM_EPI_LNB_STAT)) != (FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU | FFM_EPI_LGRAD) || rank % 4) return false;
        return true;
    }
    if (flags & FFM_EPI_LNB_STAT) return false;
    if ((flags & FFM_EPI_LNIN) && (flags & ~(FFM_EPI_RANKOP | FFM_EPI_LNIN)) != (FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU))
        return false;                                          // ln_2 folded in: the c_fc forward epilogue only
    if ((flags & FFM_EPI_ROWSTATS) && (flags & ~(FFM_EPI_RANKOP | FFM_EPI_ROWSTATS)) != (FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL))
        return false;                                          // row sums: the c_proj forward epilogue only
    switch (flags & ~(FFM_EPI_RANKOP | FFM_EPI_ROWSTATS | FFM_EPI_LNIN)) {
        case FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU:
        case FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL:
        case FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU:
        case FFM_EPI_LORA | FFM_EPI_LORA_KR: return true;
    }
    return false;
}

}  // namespace

// Cost model (both kernels are bound by the bytes a CU pulls through its texture path): rounds x operand rows
// loaded per K step.  The 128x128 kernel runs two blocks per CU, so a CU with two tiles loads 2 x 256 rows.
int ffm_panel_select(int M, int N, int K, int flags, int rank, int dtype, bool packed) {
    if (!packed || dtype != FFM_BF16 || K % 128 != 0 || K < 512) return -1;
    const bool rk = (flags & FFM_EPI_RANKOP) != 0;
    if (rk ? !rk_flags_ok(flags, rank) : !plain_flags_ok(flags)) return -1;
    // FFM_PANEL=off: always the 128x128 kernel (A/B runs); read once per process, not per launch
    static const bool panel_off = [] {
        const char* f = getenv("FFM_PANEL");
        return f && (f[0] == 'o' || f[0] == '0');
    }();
    if (panel_off) return -1;
    // Configurations 5 and up are enabled by a bit mask (default FFM_PANEL_MASK_DEFAULT; FFM_PANEL_MASK=<int> overrides, A/B
    // runs): 5 = 176x128 FairLoRA two waves per SIMD, 6 = 160x128 plain ditto, 7 = 208x384 FairLoRA ditto, 8 = 160x128
    // FairLoRA, 9 = 240x256 plain, 10 = 240x256 plain two waves per SIMD, 11 / 12 = 160x128 FairLoRA / plain with the K split
    static const int exp_mask = [] {
        const char* f = getenv("FFM_PANEL_MASK");
        return f ? atoi(f) : FFM_PANEL_MASK_DEFAULT;
    }();
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    long best = ((t128 + 255) / 256) * 256;
    int pick = -1;
    for (int c = 0; c < FFM_PANEL_NCFG; ++c) {
        const ffm_panel_cfg& cf = FFM_PANEL_CFGS[c];
        const int bm = 16 * cf.mf, bn = ffm_panel_bn(cf), nfe = bn / 64;      // nfe: the tile's width in 64-column units
        if (N % bn || cf.rankop != rk) continue;
        if (c >= 5 && !((exp_mask >> c) & 1)) continue;
        if (c == 5 || c == 6 || c == 9) continue;              // measured in round 3, lost, no longer instantiated
        if (cf.ks && K % 256) continue;                       // the K-split loop is unrolled by four K64 steps
        if ((flags & FFM_EPI_LGRAD) && c != 7) continue;      // instantiated for the 8-wave 208x384 tile only
        if ((flags & FFM_EPI_LNB_APPLY) && c != (rk ? 8 : 2)) continue;  // ... and for the 4-wave 160x128 tiles only
        if ((flags & FFM_EPI_ROWSTATS) && ((2 * cf.nf) & (2 * cf.nf - 1))) continue;   // row sums: power-of-two lanes per row
        // measured (tools/bench_panel.py): with a plain epilogue and a short K the 256-wide tile does not pay for the
        // un-overlapped prologue / store burst of a single round (qkv, K = 768: 34.6 us against 32.5 us)
        const int per_cu = cf.per_cu;
        if (!rk && nfe >= 4 && K < 1536 && per_cu == 1 && c < 5) continue;
        const long blocks = (long)((M + bm - 1) / bm) * (N / bn);
        // more than one round of tiles loses to the 128x128 kernel, whose two blocks per CU overlap one tile's epilogue
        // with the other's main loop (qkv at bs 32: 720 blocks of 160x128, 36.9 us against 32.5 us)
        // ... except the 208x384 FairLoRA tile at several rounds (bs 64: 488 blocks, 3D OCT: 19 700 rows -> 760 blocks): its epilogues
        // run at the HBM rate since round 2, and three rounds of it (~160 us) beat the 128x128 kernel's 206-228 us
        const bool multi = rk && nfe == 6 && blocks > 256;
        if (blocks > 256 * per_cu && !multi) continue;
        // (a two-per-CU tile that fills less than half of its slots is a one-per-CU tile with a worse shape)
        if (per_cu > 1 && blocks <= 256) continue;
        // (the two-waves-per-SIMD twin of a tile wins the tie)
        const long cost = 2 * (long)per_cu * (bm + bn) * (multi ? (blocks + 255) / 256 : 1) - (cf.pw == 8 ? 1 : 0);   // (pw 8: two waves per SIMD)
        if (cost < 2 * best) { best = (cost + 1) / 2; pick = c; }
    }
    return pick;
}

int ffm_panel_ds_rows(int M, int N, int cfg) {
    const int bm = 16 * FFM_PANEL_CFGS[cfg].mf, bn = ffm_panel_bn(FFM_PANEL_CFGS[cfg]);
    return ((M + bm - 1) / bm) * (N / bn);
}

int ffm_panel_tiles_n(int N, int cfg) { return N / ffm_panel_bn(FFM_PANEL_CFGS[cfg]); }

#define PANEL_CASE(F)                                                                      \
    case F:                                                                                \
        switch (cfg) {                                                                     \
            case 1: return ffm_panel::launch_panel<16, 4, false, F>(a, s);                 \
            case 2: return ffm_panel::launch_panel<10, 2, false, F>(a, s);                 \
            case 4: return ffm_panel::launch_panel<8, 4, false, F>(a, s);                  \
            case 10: return ffm_panel::launch_panel<15, 2, false, F, 8>(a, s);             \
            case 12: return ffm_panel::launch_panel<10, 2, false, F, 8, 1>(a, s);          \
        }                                                                                  \
        return FFM_EINVAL;

int ffm_panel_launch(const ffm_gemm_args& a, int cfg, hipStream_t s) {
    if (((uintptr_t)a.b_packed & 15) || a.ldc % 8) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_ROWSTATS) && !a.rowstat_part) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_LNIN) && (!a.ln_part || !a.ln_c || a.ln_np <= 0 || a.ln_np > 8 || !a.bias)) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_LNIN) && (a.flags & FFM_EPI_RANKOP) && !a.ln_rk) return FFM_EINVAL;
    if (a.flags & FFM_EPI_RANKOP) return ffm_panel_launch_rk(a, cfg, s);
    switch (a.flags) {
        PANEL_CASE(0)
        PANEL_CASE(FFM_EPI_BIAS)
        PANEL_CASE(FFM_EPI_BIAS | FFM_EPI_RESIDUAL)
        PANEL_CASE(FFM_EPI_BIAS | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS)
        PANEL_CASE(FFM_EPI_BIAS | FFM_EPI_LNIN)
        case FFM_EPI_LNB_APPLY:                                        // dX of the in-projection applying ln_1's backward
            if (cfg == 2) return ffm_panel::launch_panel<10, 2, false, FFM_EPI_LNB_APPLY>(a, s);
            return FFM_EINVAL;
    }
    return FFM_EINVAL;
}

extern "C" int ffm_pack_b(const void* src, void* dst, int N, int K, int ld, void* stream) {
    if (!src || !dst || N <= 0 || K <= 0 || N % 16 || K % 32 || ld < K || ld % 8) return FFM_EINVAL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return FFM_EINVAL;
    const size_t total = (size_t)(N >> 4) * (K >> 5) * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_b_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, N, K,
                       ld);
    FFM_CHECK_LAUNCH();
    return FFM_OK;
}
