// Panel GEMM, FairLoRA epilogues (FFM_EPI_RANKOP): c_fc / c_proj forward and their dX products.
// The instantiations are split over three translation units by tile configuration (gemm_panel_rk.hip: the one-wave-per-
// SIMD 208x384 and 176x128 tiles; gemm_panel_rk2.hip: the 160x128 tiles, 4 waves and the 8-wave K split;
// gemm_panel_rk3.hip: the 8-wave 208x384 tile) so that they compile side by side: one unit took 6.5 minutes.
#include "gemm_panel_impl.h"

int ffm_panel_launch_rk2(const ffm_gemm_args& a, int cfg, hipStream_t s);      // gemm_panel_rk2.hip
int ffm_panel_launch_rk3(const ffm_gemm_args& a, int cfg, hipStream_t s);      // gemm_panel_rk3.hip

#define PANEL_RK_CASE(F)                                                                   \
    case F:                                                                                \
        switch (cfg) {                                                                     \
            case 0: return ffm_panel::launch_panel<13, 6, true, F>(a, s);       \
            case 3: return ffm_panel::launch_panel<11, 2, true, F>(a, s);       \
        }                                                                                  \
        return FFM_EINVAL;

static int ffm_panel_launch_rk1(const ffm_gemm_args& a, int cfg, hipStream_t s) {
    switch (a.flags & ~FFM_EPI_RANKOP) {
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU)                          // c_fc forward
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU | FFM_EPI_LNIN)           // ... with ln_2 folded in
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL)                      // c_proj forward
        case FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS:            // ... leaving row sums for ln_1
            // (the 128-column tiles only: a row's lanes must form a power-of-two group)
            if (cfg == 3) return ffm_panel::launch_panel<11, 2, true, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS>(a, s);
            return FFM_EINVAL;
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU)                      // dX of c_proj
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR)                                      // dX of c_fc
    }
    return FFM_EINVAL;
}

int ffm_panel_launch_rk(const ffm_gemm_args& a, int cfg, hipStream_t s) {
    if (!a.rk || ((uintptr_t)a.rk & 15) || !a.S || !a.lw || a.rank <= 0 || a.rank > 16) return FFM_EINVAL;
    if ((a.flags & FFM_EPI_LGRAD) && (!a.lg_v || ((uintptr_t)a.lg_v & 15) || !a.lg_part_c || !a.lg_part_a || a.rank % 4 || a.gelu_deriv || cfg != 7))
        return FFM_EINVAL;
    switch (cfg) {
        case 0: case 3: return ffm_panel_launch_rk1(a, cfg, s);
        case 8: case 11: return ffm_panel_launch_rk2(a, cfg, s);
        case 7: return ffm_panel_launch_rk3(a, cfg, s);
    }
    return FFM_EINVAL;
}
