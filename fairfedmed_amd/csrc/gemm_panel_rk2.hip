// Panel GEMM, FairLoRA epilogues: the 160x128 tiles (configuration 8: 4 waves; 11: 8 waves as 4 column slabs x 2 K halves).
// See gemm_panel_rk.hip.
#include "gemm_panel_impl.h"

#define PANEL_RK_CASE(F)                                                                   \
    case F:                                                                                \
        switch (cfg) {                                                                     \
            case 8: return ffm_panel::launch_panel<10, 2, true, F>(a, s);       \
            case 11: return ffm_panel::launch_panel<10, 2, true, F, 8, 1>(a, s); \
        }                                                                                  \
        return FFM_EINVAL;

int ffm_panel_launch_rk2(const ffm_gemm_args& a, int cfg, hipStream_t s) {
    switch (a.flags & ~FFM_EPI_RANKOP) {
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU)                          // c_fc forward
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU | FFM_EPI_LNIN)           // ... with ln_2 folded in
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL)                      // c_proj forward
        case FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS:            // ... leaving row sums for ln_1
            // (the 128-column tiles only: a row's lanes must form a power-of-two group)
            if (cfg == 8) return ffm_panel::launch_panel<10, 2, true, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS>(a, s);
            if (cfg == 11) return ffm_panel::launch_panel<10, 2, true, FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS, 8, 1>(a, s);
            return FFM_EINVAL;
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU)                      // dX of c_proj
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR)                                      // dX of c_fc
        case FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_LNB_APPLY:                           // ... applying ln_2's backward
            if (cfg == 8) return ffm_panel::launch_panel<10, 2, true, FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_LNB_APPLY>(a, s);
            return FFM_EINVAL;
    }
    return FFM_EINVAL;
}
