// Panel GEMM, FairLoRA epilogues: the 208x384 tile with two waves per SIMD (configuration 7).  See gemm_panel_rk.hip.
#include "gemm_panel_impl.h"

#define PANEL_RK_CASE(F)                                                                   \
    case F:                                                                                \
        switch (cfg) {                                                                     \
            case 7: return ffm_panel::launch_panel<13, 3, true, F, 8>(a, s);    \
        }                                                                                  \
        return FFM_EINVAL;

int ffm_panel_launch_rk3(const ffm_gemm_args& a, int cfg, hipStream_t s) {
    switch (a.flags & ~FFM_EPI_RANKOP) {
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU)                          // c_fc forward
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_GELU | FFM_EPI_LNIN)           // ... with ln_2 folded in
        PANEL_RK_CASE(FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL)                      // c_proj forward
        case FFM_EPI_BIAS | FFM_EPI_LORA | FFM_EPI_RESIDUAL | FFM_EPI_ROWSTATS:            // ... leaving row sums for ln_1
            // (the 128-column tiles only: a row's lanes must form a power-of-two group)
            return FFM_EINVAL;
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU)                      // dX of c_proj
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU | FFM_EPI_LGRAD)      // ... with the two gradient partial products
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR | FFM_EPI_DGELU | FFM_EPI_LGRAD | FFM_EPI_LNB_STAT)   // ... and ln_2's backward row sums
        PANEL_RK_CASE(FFM_EPI_LORA | FFM_EPI_LORA_KR)                                      // dX of c_fc
    }
    return FFM_EINVAL;
}
