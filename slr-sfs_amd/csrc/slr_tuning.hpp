// slr_tuning.hpp -- the compile-time constants of libslrsplat that a measurement chose between, in one place, each with the value that
// ships and the numbers behind it (MI355X, gfx950).  Variants that were measured and REJECTED are not in the sources any more: their numbers
// are in HISTORY.md with the commit that holds the code.  A variant build re-defines constants through one hook:
//     make -C slr-sfs_amd/csrc OUT=../lib/var_x.so TUNE="SLR_XCD_GROUP=8 SLR_LMAX=32"
// (the Makefile writes the #undef / #define pairs into a header this file includes last) and is compared with SLR_SFS_AMD_LIB=<variant>
// (tools/dropin_bench.py, tools/small_grid_bench.py, tools/dev/variants.sh).
#pragma once

// ---- geometry (slr_common.hpp)
#define SLR_TILE_H 8            // output tile = 8 rows x 64 columns = 512 work-items.  4: incoherent +28 %; 16: +6..15 %; 1 / 2: 1.5-3x slower
#define SLR_EPT_ONE 2           // bin entries per work-item, one flow: segment = 1024 entries (46 KiB of LDS with SLR_REC6 + SLR_CHUNK_ONE 4)
#define SLR_EPT_TWO 3           // two flows (fused frame): segment = 1536 entries (66 KiB, two workgroups per CU).  2: almost every tile becomes
                                // multi-segment, a frame goes from 340 to 610 us

// ---- tile kernel
#define SLR_STORE_AUX 2         // cache policy of the output stores (buffer instruction aux bits on gfx950: 1 sc0, 2 nt, 16 sc1).  Round 5, fused clip kernel, us per
                                // frame on one box: default policy 154.3 / 155.3, nt 150.5, sc1 154.8, sc0 + nt 150.2 (the kernel never reads its output: nt keeps it
                                // out of the feature planes' way in L2)
#define SLR_WAVES_SCAN 4        // scan tile kernel: <= 128 VGPRs (it uses ~100) = two workgroups per CU, which is what its 63 KiB of LDS allow; 5 (<= 102): 8 bytes of scratch in the normalising variant
#define SLR_KREG_TWO 6          // ... two flows
#define SLR_LMAX 16             // records a work-item walks alone before the wave helps (8 / 16 / 32: within 1 %)
#define SLR_HEAVY_SLACK 24      // a list this much longer than the wave's share is walked by the whole wave (8: t=59 +55 %; 64: +10 %)
#define SLR_HEAVY_MAX 4         // at most this many cooperative (whole-wave) list walks per wave and chunk; more long lists: every lane walks its own
#define SLR_BAL_SLACK 24        // scan kernels: a tile whose longest record list exceeds its positions per work-item by more than this takes the balanced gather (splat_tile.hpp)
#define SLR_XCD_GROUP 4         // neighbouring tiles kept on one XCD (column halo from its L2: -12 % HBM fetch).  1 / 2 / 4 / 8: 183.9 / 182.9 / 185.5 / 182.3 us per frame
#define SLR_BATCH_INTERLEAVE 1  // block groups of the frames of a launch dealt round-robin: same tile of consecutive frames shares an L2 (fetch 770 -> 482 MB per frame)

// ---- plan (bins front end)
#define SLR_PLAN_HEAVY 6        // single launches: tiles with more than 6/4 of the mean entry count go first (t=30 149.6 -> 142.4 us; 5/4 .. 8/4 within 2 %)

// ---- small grids / scan front end
#define SLR_CSPLIT_MAX 4        // channel groups per tile on grids smaller than the chip (256x480: 2 groups 37.5 -> 33.5 us; 128x240: 4 groups 34 -> 21 us)
#define SLR_CSPLIT_SLOTS 512    // workgroup slots of the chip the groups may fill (256 CUs x 2).  1024 / 8 groups: config C2 39 -> 54 us
#define SLR_SCAN_DEFER_AT 896    // scan tile kernel: a tile of more entries than this goes to the sink launch (<= 1024 = one segment; its near-full tiles are the
                                // tile kernel's tail: 8 chunks at 3 us).  1024 / 896 / 768 / 640, us: C2 grid t=30 82 / 80 / 84 / 84, training shape t=30 91 / 89 / 87 / 86,
                                // t=59 130 / 117 / 120 / 121, 384x640 t=30 106 / 106 / 119 / 126, t=59 135 / 139 / 131 / 133; incoherent flows: no tile above 640
#define SLR_SINK_PIECES 33      // sink launch of the scan front end (splat_op.hip: op_sink_kernel): piece slots = emergency slabs of pieces that find the pool empty; the task list takes min(32, this) pieces per round ...
#define SLR_SINK_TASKS 16       // ... x task slots per piece (a task = 2 candidate source tiles = 16 row segments <= 1024 entries) ...
#define SLR_SINK_POOL_MB 64     // ... bytes of slabs in the workspace (a slab = the partial sums of one task slot: (planes of its channel group + 1) x 2 KiB)
#define SLR_SINK_ENT_MB 16      // ... bytes of entries (16 each) the deferred pieces of a call may write out; pieces beyond that are cut by candidate pairs
#define SLR_SINK_GROUPS 8       // ... x channel groups
#define SLR_SINK_GRID 512       // ... workgroups of the sink launch: what the chip holds of them at once (256 CUs x 2, a multiple of 8); each takes the tasks lin, lin + grid, ... of the call's ordered task list
#define SLR_SCAN_MAX_TILES 1024 // default of slr_splat_set_scan_max_tiles: one-flow calls on grids of at most this many tiles take the scan front end,
                                // larger ones the rows front end (config C2, 240 tiles: bins 56 / scan 39 / rows 56 us; 384x640, 960 tiles: incoherent
                                // 80 / 61 / 77, Euler t=30 124 / 116 / 112; 768x1280, 1920 tiles: identity 175 / 142 / 146, t=30 202 / 207 / 165, t=59 245 / 278 / 216)

// ---- rows front end (row segments binned per tile, plan in the same launch)
#define SLR_ROW_CAP 256         // row segments (64 source pixels of one image row) a tile's list holds; a tile touched by more is
                                // scanned from the whole flow instead (pathological flows only; identity ~30, Euler t=59 < 200)
#define SLR_ROWBIN_R 2          // source tiles (vertically adjacent) per workgroup of rowbin_kernel = row segments per wave.  1 / 2 / 4 (whole call, us): identity 153 / 148 / 148, Euler t=30 177 / 173 / 180, t=59 225 / 221 / 233
#define SLR_CLIP_MAXB 16        // frames per launch of the fused clip kernel (kernel arguments: 16 x 112 bytes)
#define SLR_CLIP_ALIGNED 0      // clip plans: 1 = the first piece of tile t is item t in every frame (further pieces behind item nt - 1)
#define SLR_CLIP_HEAVY 0        // clip plans: tiles with more than 7/8 of a segment's entries (and every tile cut into pieces) go first; 0 = one row-major pass (measured: 7/8 161.4-162.1 us per frame, 6/8 162.1-162.7, one pass 159.5-160.4: the spatial order is worth more than the shorter tail)
#define SLR_WAVES_CLIP 4         // waves per SIMD the fused clip kernel is compiled for (86 VGPRs as built; its 79 KiB of LDS allow two workgroups per CU)
#define SLR_ROWBIN_CLIP_R 4     // rowbin_clip_kernel: image rows per wave (their flow loads in flight together, their appends in one flush).  Stage us per frame, same box: 2: 172.0-172.8, 3: 171.7, 4: 171.4, 6: 171.1
#define SLR_ROW_CB_CLIP 4       // fused clip kernel: row segments per group of the walk (two groups' flow loads in flight: a tile of two flows has ~60 segments, 8 per wave)
#define SLR_ROW_CB 3            // row segments per wave whose flow loads are in flight together (2 / 3 / 4: within 1 %; 4 needs 3 more registers)
#define SLR_WAVES_ROWS 6        // waves per SIMD the rows tile kernel is compiled for: 80 VGPRs = three workgroups per CU (two: +5..10 %)
#define SLR_KREG_ROWS 4         // register-resident records per output pixel in the rows tile kernel 
#define SLR_ROWS_PLAN_PER 4     // tiles per work-item and round of the plan (the plan's registers set rowbin_kernel's occupancy)
#define SLR_ROWS_FILL 7         // plan: a piece of a heavy tile may hold up to this many eighths of a segment by the (estimated) histogram.  6 / 5 / 4: Euler t=59 +0 / +2 / +8 %, t=45 +2 / +5 / +11 % (more, smaller pieces lose: every piece repeats the tile's list, sort and scan)
#define SLR_EPT_DEFER 4         // entries per work-item and pass in the pass-by-pass launch of the rows front end (4: passes of 2048 entries, 86 KiB of LDS)
#define SLR_ROWS_GROUP 2        // narrow pieces: 0 one lane per output pixel; 1 groups of 8 / 4 lanes up to 16 columns; 2 also pairs up to 32 columns
#define SLR_FRONT_END -1        // default of slr_splat_set_front_end: -1 = by grid size, 0 bins, 1 scan (boxes), 2 rows

#define SLR_CONV_SKIP_FILL 3     // 3x3 kernels with the block's 1x1 skip inside (conv.hip, SKIP): skip chunks staged per barrier pair (16 KiB of LDS and 16 registers in flight each)

#define SLR_CONV_DEEP_STAGE 1    // 3x3 split kernels with fewer than 8 rows per wave: staging loads five taps ahead of their use (three register sets) instead of two

// ---- development aids

// ---- variant builds (csrc/Makefile: TUNE="NAME=value ...")
#ifdef SLR_TUNING_OVERRIDE
#include SLR_TUNING_OVERRIDE
#endif
