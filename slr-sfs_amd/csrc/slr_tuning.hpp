// slr_tuning.hpp -- every compile-time knob of libslrsplat in one place, with the value that ships and the measurement
// behind it (MI355X, gfx950; details in DESIGN.md).  Variant builds override them on the command line:
//     make -C slr-sfs_amd/csrc OUT=../lib/var_x.so DEFS="-DSLR_XCD_GROUP=8"
// and are compared with SLR_SFS_AMD_LIB=<variant> (tools/dropin_bench.py, tools/frontend_bench.py, tools/dev/variants.sh).
#pragma once

// ---- geometry (slr_common.hpp)
#ifndef SLR_TILE_H
#define SLR_TILE_H 8            // output tile = 8 rows x 64 columns = 512 work-items.  4: incoherent +28 %; 16: +6..15 %; 1 / 2: 1.5-3x slower
#endif
#ifndef SLR_EPT_ONE
#define SLR_EPT_ONE 2           // bin entries per work-item, one flow: segment = 1024 entries (46 KiB of LDS with SLR_REC6 + SLR_CHUNK_ONE 4)
#endif
#ifndef SLR_EPT_TWO
#define SLR_EPT_TWO 3           // two flows (fused frame): segment = 1536 entries (66 KiB, two workgroups per CU).  2: almost every tile becomes
#endif                          // multi-segment, a frame goes from 340 to 610 us
#ifndef SLR_EPT_SCAN
#define SLR_EPT_SCAN 2          // SCAN instantiation.  3 (fewer shared tiles): 768x1280 Euler t=30 206 -> 188 us, but identity 145 -> 152, config C2 39 -> 44
#endif

// ---- tile kernel
#ifndef SLR_STORE_AUX
#define SLR_STORE_AUX 2         // cache policy of the output stores (buffer instruction aux bits on gfx950: 1 sc0, 2 nt, 16 sc1).  Round 5, fused clip kernel, us per
#endif                          // frame on one box: default policy 154.3 / 155.3, nt 150.5, sc1 154.8, sc0 + nt 150.2 (the kernel never reads its output: nt keeps it
                                // out of the feature planes' way in L2)
#ifndef SLR_REC_SORT
#define SLR_REC_SORT 0          // 1: the register-resident records of an output pixel sorted by staged entry (splat_tile.hpp: pixel_list).  Round 5: no gain
#endif                          // (159.9 sorted / 158.2 unsorted us per frame on the same box): the gather is not where the time is (see SLR_SKIP)
#ifndef SLR_PK_FMA
#define SLR_PK_FMA 0            // 1: the gather's FMAs as v_pk_fma_f32 with the weight broadcast from the LOW register of the record's pair (splat_tile.hpp: accum4):
#endif                          // -14 % VALU instructions per frame, no gain in time (152.7 vs 151.3 us per frame): the kernel is not VALU-bound
#ifndef SLR_CHUNK_ONE
#define SLR_CHUNK_ONE 4         // planes staged per pass, one flow.  8 (round 1): 152 / 196 / 244 / 228 us vs 144 / 178 / 223 / 208 (identity / t=30 / t=59 / incoherent)
#endif
#ifndef SLR_REC6
#define SLR_REC6 1              // one-flow / scan variants: records as (u16 entry, f32 weight), 6 instead of 8 bytes.  The two-flow variant keeps
#endif                          // 8-byte records (one ds_read_b64 per record: 201 vs 206 us per frame)
#ifndef SLR_WAVES_ONE
#define SLR_WAVES_ONE 5         // __launch_bounds__ waves per SIMD, one-flow: <= 96 VGPRs.  4 / 5 / 6 measure the same (118.7 / 119.2 / 118.2 us identity);
#endif                          // 6 leaves the normalising variant two registers short
#ifndef SLR_WAVES_SCAN
#define SLR_WAVES_SCAN 5        // scan tile kernel: 83 VGPRs, two workgroups per CU, no scratch (6: 80 VGPRs + 12 bytes of scratch, three per CU -- the same time on every small grid)
#endif
#ifndef SLR_KREG_ONE
#define SLR_KREG_ONE 4          // records of an output pixel kept in registers across the chunks (one flow; 8 / 10: < 1 % gain)
#endif
#ifndef SLR_KREG_TWO
#define SLR_KREG_TWO 6          // ... two flows
#endif
#ifndef SLR_KREG_SCAN
#define SLR_KREG_SCAN 4         // ... SCAN instantiation (3 saves four registers, not needed)
#endif
#ifndef SLR_LMAX
#define SLR_LMAX 16             // records a work-item walks alone before the wave helps (8 / 16 / 32: within 1 %)
#endif
#ifndef SLR_HEAVY_SLACK
#define SLR_HEAVY_SLACK 24      // a list this much longer than the wave's share is walked by the whole wave (8: t=59 +55 %; 64: +10 %)
#endif
#ifndef SLR_HEAVY_MAX
#define SLR_HEAVY_MAX 4         // at most this many cooperative (whole-wave) list walks per wave and chunk; more long lists: every lane walks its own
#endif
#ifndef SLR_XCD_GROUP
#define SLR_XCD_GROUP 4         // neighbouring tiles kept on one XCD (column halo from its L2: -12 % HBM fetch).  1 / 2 / 4 / 8: 183.9 / 182.9 / 185.5 / 182.3 us per frame
#endif
#ifndef SLR_MAXB
#define SLR_MAXB 8              // frames per launch of the fused kernel (kernel arguments: 8 x 256 bytes).  1 / 4 / 8: 239 / 208 / 195 us per frame of work
#endif
#ifndef SLR_BATCH_INTERLEAVE
#define SLR_BATCH_INTERLEAVE 1  // block groups of the frames of a launch dealt round-robin: same tile of consecutive frames shares an L2 (fetch 770 -> 482 MB per frame)
#endif

// ---- plan (bins front end)
#ifndef SLR_PLAN_SY
#define SLR_PLAN_SY 1           // tile rows per super-tile of the work-item order.  2 (+ XCD group 8): -7 % fetch, +1 % time
#endif
#ifndef SLR_PLAN_HEAVY
#define SLR_PLAN_HEAVY 6        // single launches: tiles with more than 6/4 of the mean entry count go first (t=30 149.6 -> 142.4 us; 5/4 .. 8/4 within 2 %)
#endif

// ---- small grids / scan front end
#ifndef SLR_CSPLIT_MAX
#define SLR_CSPLIT_MAX 4        // channel groups per tile on grids smaller than the chip (256x480: 2 groups 37.5 -> 33.5 us; 128x240: 4 groups 34 -> 21 us)
#endif
#ifndef SLR_CSPLIT_SLOTS
#define SLR_CSPLIT_SLOTS 512    // workgroup slots of the chip the groups may fill (256 CUs x 2).  1024 / 8 groups: config C2 39 -> 54 us
#endif
#ifndef SLR_SINK_PIECES
#define SLR_SINK_PIECES 33      // sink launch of the scan front end (splat_op.hip: op_sink_kernel): deferred pieces rendered at once (more: the list is looped) ...
#endif
#ifndef SLR_SINK_TASKS
#define SLR_SINK_TASKS 16       // ... x task slots per piece (a task = 2 candidate source tiles = 16 row segments <= 1024 entries) ...
#endif
#ifndef SLR_SINK_ORDER
#define SLR_SINK_ORDER 0        // grid of the sink launch: 0 = pieces x groups x task slots (with an ODD number of piece slots a piece's workgroups land on different XCDs), 1 = task slots x groups x pieces
#endif
#ifndef SLR_SINK_POOL_MB
#define SLR_SINK_POOL_MB 64     // ... bytes of slabs in the workspace (a slab = the partial sums of one task slot: (planes of its channel group + 1) x 2 KiB)
#endif
#ifndef SLR_SINK_ENT_MB
#define SLR_SINK_ENT_MB 16      // ... bytes of entries (16 each) the deferred pieces of a call may write out; pieces beyond that are cut by candidate pairs
#endif
#ifndef SLR_SINK_GROUPS
#define SLR_SINK_GROUPS 8       // ... x channel groups
#endif
#ifndef SLR_SCAN_MAX_TILES
#define SLR_SCAN_MAX_TILES 1024 // default of slr_splat_set_scan_max_tiles: one-flow calls on grids of at most this many tiles take the scan front end,
#endif                          // larger ones the rows front end (config C2, 240 tiles: bins 56 / scan 39 / rows 56 us; 384x640, 960 tiles: incoherent
                                // 80 / 61 / 77, Euler t=30 124 / 116 / 112; 768x1280, 1920 tiles: identity 175 / 142 / 146, t=30 202 / 207 / 165, t=59 245 / 278 / 216)
#ifndef SLR_SCAN_CB
#define SLR_SCAN_CB 5           // candidate source tiles per group of the scan (two groups' flow loads in flight).  3 / 4 / 5 / 8: C2 36.7 / 38.6 / 38.6 / 40.8 us
#endif
#ifndef SLR_SCAN_SHARE
#define SLR_SCAN_SHARE 1        // heavy tiles share their segments through the work queue.  0 (one workgroup walks them): 768x1280 t=30 236 / t=59 336 us vs 212 / 280
#endif
#ifndef SLR_SHARE_HELPERS
#define SLR_SHARE_HELPERS 16    // one workgroup in this many looks for shared segments after its own tile.  4 / 8 / 16: t=30 220 / 219 / 212 us, t=59 309 / 292 / 280
#endif
#ifndef SLR_SHARE_STORE
#define SLR_SHARE_STORE 1       // partial slots: 0 = sc0 sc1 stores, 1 = sc1 stores, 2 = plain stores + an agent release fence per wave (all within 3 %)
#endif

// ---- rows front end (row segments binned per tile, plan in the same launch)
#ifndef SLR_ROW_CAP
#define SLR_ROW_CAP 256         // row segments (64 source pixels of one image row) a tile's list holds; a tile touched by more is
#endif                          // scanned from the whole flow instead (pathological flows only; identity ~30, Euler t=59 < 200)
#ifndef SLR_ROWBIN_R
#define SLR_ROWBIN_R 2          // source tiles (vertically adjacent) per workgroup of rowbin_kernel = row segments per wave.  1 / 2 / 4 (whole call, us): identity 153 / 148 / 148, Euler t=30 177 / 173 / 180, t=59 225 / 221 / 233
#endif
#ifndef SLR_PREFETCH_BURST_ONE
#define SLR_PREFETCH_BURST_ONE 0 // one-flow kernels: 1 = the plane loads of the chunk after next in one burst right behind the barrier; 0 = one plane at each stop of the gather
#endif
#ifndef SLR_CLIP_MAXB
#define SLR_CLIP_MAXB 16        // frames per launch of the fused clip kernel (kernel arguments: 16 x 112 bytes)
#endif
#ifndef SLR_CLIP_ALIGNED
#define SLR_CLIP_ALIGNED 0      // clip plans: 1 = the first piece of tile t is item t in every frame (further pieces behind item nt - 1)
#endif
#ifndef SLR_CLIP_HEAVY
#define SLR_CLIP_HEAVY 0        // clip plans: tiles with more than 7/8 of a segment's entries (and every tile cut into pieces) go first; 0 = one row-major pass (measured: 7/8 161.4-162.1 us per frame, 6/8 162.1-162.7, one pass 159.5-160.4: the spatial order is worth more than the shorter tail)
#endif
#ifndef SLR_WAVES_CLIP
#define SLR_WAVES_CLIP 4         // waves per SIMD the fused clip kernel is compiled for (86 VGPRs as built; its 79 KiB of LDS allow two workgroups per CU)
#endif
#ifndef SLR_ROWBIN_CLIP_R
#define SLR_ROWBIN_CLIP_R 4     // rowbin_clip_kernel: image rows per wave (their flow loads in flight together, their appends in one flush).  Stage us per frame, same box: 2: 172.0-172.8, 3: 171.7, 4: 171.4, 6: 171.1
#endif
#ifndef SLR_ROW_CB_CLIP
#define SLR_ROW_CB_CLIP 4       // fused clip kernel: row segments per group of the walk (two groups' flow loads in flight: a tile of two flows has ~60 segments, 8 per wave)
#endif
#ifndef SLR_ROW_CB
#define SLR_ROW_CB 3            // row segments per wave whose flow loads are in flight together (2 / 3 / 4: within 1 %; 4 needs 3 more registers)
#endif
#ifndef SLR_WAVES_ROWS
#define SLR_WAVES_ROWS 6        // waves per SIMD the rows tile kernel is compiled for: 80 VGPRs = three workgroups per CU (two: +5..10 %)
#endif
#ifndef SLR_KREG_ROWS
#define SLR_KREG_ROWS 4         // register-resident records per output pixel in the rows tile kernel 
#endif
#ifndef SLR_ROWS_PLAN_PER
#define SLR_ROWS_PLAN_PER 4     // tiles per work-item and round of the plan (the plan's registers set rowbin_kernel's occupancy)
#endif
#ifndef SLR_ROWS_FILL
#define SLR_ROWS_FILL 7         // plan: a piece of a heavy tile may hold up to this many eighths of a segment by the (estimated) histogram.  6 / 5 / 4: Euler t=59 +0 / +2 / +8 %, t=45 +2 / +5 / +11 % (more, smaller pieces lose: every piece repeats the tile's list, sort and scan)
#endif
#ifndef SLR_EPT_DEFER
#define SLR_EPT_DEFER 4         // entries per work-item and pass in the pass-by-pass launch of the rows front end (4: passes of 2048 entries, 86 KiB of LDS)
#endif
#ifndef SLR_ROWS_EVEN_FIRST
#define SLR_ROWS_EVEN_FIRST 0   // plan: 1 = plain halves / quarters of a heavy tile where the histogram says they fit, the greedy cut otherwise (measured: t=30 166.0 vs 166.5, t=59 210 vs 215 us)
#endif
#ifndef SLR_ROWS_GROUP
#define SLR_ROWS_GROUP 2        // narrow pieces: 0 one lane per output pixel; 1 groups of 8 / 4 lanes up to 16 columns; 2 also pairs up to 32 columns
#endif
#ifndef SLR_ROW_SORT
#define SLR_ROW_SORT 1          // the tile kernel puts its row-segment list into image order before scanning (the appends arrive in any order).  0: identity 148 -> 157 us, Euler t=30 173 -> 192, t=59 221 -> 249
#endif
#ifndef SLR_FRONT_END
#define SLR_FRONT_END -1        // default of slr_splat_set_front_end: -1 = by grid size, 0 bins, 1 scan (boxes), 2 rows
#endif

// ---- development aids
#ifndef SLR_OUT_B8_EXP
#define SLR_OUT_B8_EXP 0          // experiment: the clip kernels on blocked planes also WRITE channel-blocked by 8 (tools/dev/b4_check.py --out-b8): correct,
                                 // and 209 against 135 us per frame -- a chunk's 16-byte store fills HALF of a pixel's 32 bytes, the other half comes a chunk later
#endif
#ifndef SLR_SKIP
#define SLR_SKIP 0              // deletion experiments on the chunk pipeline (WRONG results; timing only): 1 no plane loads, 2 no staging stores,
#endif                          // 4 no register-record reads, 8 no list loop, 16 no output stores (first chunk excepted), 32 no barriers
#ifndef SLR_DBG
#define SLR_DBG 0               // 1 drain vmcnt before staging, 4 verify staged values against global memory
#endif
#ifndef SLR_LDS_PAD
#define SLR_LDS_PAD 0           // bytes of unused LDS in front of the staged values
#endif
