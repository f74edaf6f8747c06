// Internal interface between gemm.hip (dispatcher, 128x128 kernel) and gemm_panel*.hip (panel kernel).
#pragma once
#include "common.h"

// Panel-kernel tile configurations: block tile (16*MF) x (16*PW*NF), PW waves side by side along N, 16*NF columns each.
// PW = 4: one wave per SIMD (up to 312 accumulator registers per wave); PW = 8: two waves per SIMD (<= 256 registers).
// With PW = 4, MF is chosen so that the A-ring pieces, 2*MF (+2 for the rank rows under FFM_EPI_RANKOP), split evenly
// over the waves; with PW = 8 the waves that run out of pieces repeat one (gemm_panel_impl.h, PanelGeom).
struct ffm_panel_cfg {
    int mf, nf;
    bool rankop;     // instantiated for the FFM_EPI_RANKOP epilogues (true) or for the plain ones (false)
    int per_cu;      // blocks that share a CU (registers + LDS): a round is 256 * per_cu blocks
    int pw;          // waves per block
    int ks;          // 1: the 8 waves are 4 column slabs x 2 K halves (gemm_panel_impl.h, KS) - the tile is 64*nf wide
};
constexpr int FFM_PANEL_NCFG = 13;
constexpr ffm_panel_cfg FFM_PANEL_CFGS[FFM_PANEL_NCFG] = {{13, 6, true, 1, 4, 0}, {16, 4, false, 1, 4, 0}, {10, 2, false, 1, 4, 0}, {11, 2, true, 1, 4, 0},
                                                          {8, 4, false, 2, 4, 0},
                                                          // two waves per SIMD: the same tiles as 3, 2 (5, 6: measured, lost,
                                                          // not instantiated any more) and 0
                                                          {11, 1, true, 1, 8, 0}, {10, 1, false, 1, 8, 0}, {13, 3, true, 1, 8, 0},
                                                          // 8: the 160-row FairLoRA tile for N = 768 (240 blocks at 6304
                                                          // rows where the 176-row tile launches 216); 9 / 10: 240 x 256
                                                          // for qkv (243 blocks), one (9: lost, not instantiated) and two
                                                          // waves per SIMD
                                                          {10, 2, true, 1, 4, 0}, {15, 4, false, 1, 4, 0}, {15, 2, false, 1, 8, 0},
                                                          // 11 / 12: the 160x128 tiles (FairLoRA / plain) with 8 waves as
                                                          // 4 column slabs x 2 K halves (K % 256 == 0)
                                                          {10, 2, true, 1, 8, 1}, {10, 2, false, 1, 8, 1}};
constexpr int ffm_panel_bn(const ffm_panel_cfg& c) { return 16 * (c.ks ? c.pw / 2 : c.pw) * c.nf; }

// -1: use the 128x128 kernel; otherwise the index into FFM_PANEL_CFGS
int ffm_panel_select(int M, int N, int K, int flags, int rank, int dtype, bool packed);
// rows of dS partials the chosen kernel writes (tiles_m x tiles_n: every block owns a slice of its tile row)
int ffm_panel_ds_rows(int M, int N, int cfg);
int ffm_panel_tiles_n(int N, int cfg);      // column tiles (rows of rowstat_part under FFM_EPI_ROWSTATS)
int ffm_panel_launch(const ffm_gemm_args& a, int cfg, hipStream_t s);
int ffm_panel_launch_rk(const ffm_gemm_args& a, int cfg, hipStream_t s);      // gemm_panel_rk.hip

// gemm_skinny.hip: M <= 64 rows (text tower), bf16 / f32, plain epilogues
bool ffm_skinny_ok(const ffm_gemm_args& a, int dtype);
int ffm_skinny_launch(const ffm_gemm_args& a, int dtype, hipStream_t s);
