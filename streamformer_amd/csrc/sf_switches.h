// Every environment switch of the library in ONE table, read once (first use; sf_reload_switches() re-reads, which is how the
// tests flip a switch inside a process).  A/B and tuning knobs of the measurements in DESIGN.md; none of them changes results
// beyond the documented A/B (result-discarding lab switches exist only in -DSF_LAB builds, SF_LAB_SWITCH in sf_common.h).
// sf_sw(SW_X) returns the variable's value (a C string) or nullptr when unset — the semantics getenv had at the call site.
#pragma once

#define SF_SWITCH_TABLE(X) \
  X(ACC_TWO_PLANES, "SF_ACC_TWO_PLANES", "accurate mode: residual stream as two planes instead of three (A/B; max-abs 1.4e-4 instead of 5e-5)") \
  X(ASSUME_CUS, "SF_ASSUME_CUS", "size persistent grids for this many CUs (CU-masked stream experiments)") \
  X(DISABLE_ACC_FOLD, "SF_DISABLE_ACC_FOLD", "accurate mode: fp32 residual + standalone LayerNorm instead of the folded plane form (A/B)") \
  X(DISABLE_G256_SPLIT, "SF_DISABLE_G256_SPLIT", "accurate mode: keep the bf16x3 GEMMs off the persistent 256-column kernel") \
  X(DISABLE_GEMM_COLSPLIT, "SF_DISABLE_GEMM_COLSPLIT", "N = 256 j + 128 (so400m widths): the whole GEMM on 128^2 tiles instead of 256-column kernel + a 128-column tail (A/B)") \
  X(DISABLE_GEMM_MID, "SF_DISABLE_GEMM_MID", "skinny family: no 64 x 64 tiles above 512 rows") \
  X(DISABLE_GEMM_TILE, "SF_DISABLE_GEMM_TILE", "one / two clips per call: no tile-GEMM family") \
  X(DISABLE_LN_FOLD, "SF_DISABLE_LN_FOLD", "bf16 mode: standalone LayerNorm launches instead of the fold into the consumer GEMMs (A/B)") \
  X(DISABLE_RESID_PLANES, "SF_DISABLE_RESID_PLANES", "bf16 mode: fp32 residual stream instead of hi + lo planes (A/B)") \
  X(DISABLE_SKG_UNROLL, "SF_DISABLE_SKG_UNROLL", "streaming: K-parallel skinny GEMM with a run-time K-tile count (A/B)") \
  X(DISABLE_SKINNY, "SF_DISABLE_SKINNY", "send small-M GEMMs to the generic 128^2 kernel") \
  X(DISABLE_SPATIAL_DMA, "SF_DISABLE_SPATIAL_DMA", "spatial attention on the register-staged kernel") \
  X(DISABLE_SPATIAL_DMA_ACC, "SF_DISABLE_SPATIAL_DMA_ACC", "accurate mode: spatial attention on fp32 q / k / v (test coverage of that path)") \
  X(DISABLE_SPATIAL_NTC, "SF_DISABLE_SPATIAL_NTC", "spatial attention with a run-time tile count (A/B)") \
  X(DISABLE_STREAM_FOLD, "SF_DISABLE_STREAM_FOLD", "streaming: standalone LayerNorm launches (A/B)") \
  X(DISABLE_STREAM_GRAPH, "SF_DISABLE_STREAM_GRAPH", "streaming: eager launches instead of hipGraph replay") \
  X(DISABLE_TEMPORAL_DECODE, "SF_DISABLE_TEMPORAL_DECODE", "streaming: general temporal kernel instead of the single-query one") \
  X(TBWD_OWN_CU, "SF_TBWD_OWN_CU", "training: temporal attention backward as 12-wave workgroups that own a CU instead of 4-wave workgroups with their exact LDS (bit-identical, 65 against 51 us per launch; for jobs that share a device, DESIGN.md 4)") \
  X(TEMPORAL_DECODE_LANE_KEY, "SF_TEMPORAL_DECODE_LANE_KEY", "streaming: single-query temporal kernel with one key per lane (A/B against whole-line loads)") \
  X(DISABLE_TEMPORAL_DMA, "SF_DISABLE_TEMPORAL_DMA", "temporal attention on the register-staged kernel") \
  X(DISABLE_TEMPORAL_DMA_ACC, "SF_DISABLE_TEMPORAL_DMA_ACC", "accurate mode: temporal attention on fp32 q / k / v (test coverage of that path)") \
  X(EMBED_VIA_GEMM128, "SF_EMBED_VIA_GEMM128", "embedding GEMM on the generic kernel (A/B)") \
  X(G256_FORCE_BM, "SF_G256_FORCE_BM", "256-column kernel: force the row-tile height (tests of the short tiles)") \
  X(G256_NO_BM224, "SF_G256_NO_BM224", "256-column kernel: no 224-row tiles") \
  X(G256_NO_SHORT_BM, "SF_G256_NO_SHORT_BM", "256-column kernel, accurate mode: no 160 / 192-row tiles") \
  X(G256_SPLIT_MIN_N, "SF_G256_SPLIT_MIN_N", "accurate mode: smallest N the persistent kernel takes") \
  X(G256_WALK, "SF_G256_WALK", "256-column kernel: column-group width (tiles) of the column-group-major tile walk, 0 = row-major (A/B of the fabric traffic)") \
  X(G256_STORE_WT, "SF_G256_STORE_WT", "256-column kernel: bf16 outputs leave with write-through (sc1) stores that do not stay in the XCD's L2 (A/B)") \
  X(G256_STORE_NT, "SF_G256_STORE_NT", "256-column kernel: bf16 outputs leave with non-temporal stores (A/B: do the residual planes then survive in the Infinity Cache?)") \
  X(G256_STAGGER_GROUPS, "SF_G256_STAGGER_GROUPS", "256-column kernel: phase-stagger groups per XCD") \
  X(G256_STAGGER_NS, "SF_G256_STAGGER_NS", "256-column kernel: stagger step in ns (overrides the percentage)") \
  X(G256_STAGGER_ONLY, "SF_G256_STAGGER_ONLY", "256-column kernel: stagger only some launches (A/B)") \
  X(G256_STAGGER_PCT, "SF_G256_STAGGER_PCT", "256-column kernel: stagger step as a percentage of the tile period") \
  X(GEMM_MID_MIN_M, "SF_GEMM_MID_MIN_M", "skinny family: first M of the 64 x 64 tiles") \
  X(LN_BWD_BLOCKS, "SF_LN_BWD_BLOCKS", "LayerNorm backward: workgroups (tuning)") \
  X(POOL_SHARE_CU, "SF_POOL_SHARE_CU", "pooling head: the probe / combine kernels request only the LDS they use instead of a CU's whole 160 KB (A/B; with it other workgroups share their CUs, see DESIGN.md 4 on device sharing)") \
  X(PANEL_MIN_FILL_PCT, "SF_PANEL_MIN_FILL_PCT", "panel kernel: minimum last-round fill") \
  X(PANEL_PAD_CLAMP, "SF_PANEL_PAD_CLAMP", "panel kernel: clamp instead of zero-fill the padding rows (A/B)") \
  X(PANEL_STAGGER_NS, "SF_PANEL_STAGGER_NS", "panel kernel: phase-stagger step in ns") \
  X(SKINNY_MAX_M, "SF_SKINNY_MAX_M", "skinny family: largest M") \
  X(SKINNY_NO_KG, "SF_SKINNY_NO_KG", "skinny family: no K-parallel variant") \
  X(SKINNY_NB, "SF_SKINNY_NB", "streaming, folded skinny consumers: 1 = 32 x 32 tiles everywhere, 2 / 3 = force 32 x 64 / 32 x 96 tiles (A/B)") \
  X(SKINNY_NT, "SF_SKINNY_NT", "skinny family: non-temporal weight loads (A/B)") \
  X(SKINNY_TPS, "SF_SKINNY_TPS", "skinny family: tiles per slot (tuning)") \
  X(SPATIAL_PERS, "SF_SPATIAL_PERS", "lab library: persistent spatial attention kernel") \
  X(SPATIAL_TPW, "SF_SPATIAL_TPW", "spatial attention: query tiles per workgroup (tuning)") \
  X(STREAM_GRAPH_PER_POSITION, "SF_STREAM_GRAPH_PER_POSITION", "streaming: one graph per cache position instead of the position-free one (A/B)") \
  X(TILE_FOLD_MIN_M, "SF_TILE_FOLD_MIN_M", "tile-GEMM family: smallest M whose LayerNorm fold uses a statistics buffer and the 256^2 consumers") \
  X(TILE_MAX_M, "SF_TILE_MAX_M", "tile-GEMM family: largest M") \
  X(TILE_MIN_M, "SF_TILE_MIN_M", "tile-GEMM family: smallest M") \
  X(TILE_SHAPE, "SF_TILE_SHAPE", "tile-GEMM family: force a tile shape (tools/tile_lab.py)") \
  X(TRAIN_UNFUSED_TEMPORAL, "SF_TRAIN_UNFUSED_TEMPORAL", "training: temporal_attention.output.dense and temporal_dense as two launches instead of one fused projection (A/B)") \
  X(TRAIN_SIDE_STREAM, "SF_TRAIN_SIDE_STREAM", "training: 0 keeps the LoRA gradients on the caller stream (A/B)") \
  X(WGRAD_NSPLIT, "SF_WGRAD_NSPLIT", "weight-gradient GEMM: force the token splits") \
  X(WGRAD_SMALL_TILES, "SF_WGRAD_SMALL_TILES", "weight-gradient GEMM: 128^2 tiles only") \
  X(WGRAD_UNGROUPED, "SF_WGRAD_UNGROUPED", "training: one weight-gradient launch per projection instead of one per layer (test / A/B)") \

enum SfSw {
#define SF_SW_ENUM(id, name, what) SW_##id,
  SF_SWITCH_TABLE(SF_SW_ENUM)
#undef SF_SW_ENUM
  SW_COUNT
};
const char* sf_sw(SfSw k);
