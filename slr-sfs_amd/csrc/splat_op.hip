// splat_op.hip -- the one-flow operator: _FunctionSoftsplat / FunctionSoftsplat (4 modes) / the maximum splat as ONE gather per call.
//
// Replaces models/softsplat.py:157-202 (kernel_Softsplat_updateOutput: one thread per ELEMENT, 4 global fp32 atomicAdds each into a
// pre-zeroed output), its launcher :390-424, and the weighting / normalisation of FunctionSoftsplat :665-690.  Here the scatter is
// turned around (owner computes): the flow is shared by all C channels, so it is sorted once per call, and every output tile's
// workgroup gathers exactly the sources that land in it -- every output byte written once, never read, never zeroed, no global
// atomics on the values (splat_tile.hpp has the two phases every tile kernel shares).  Two exact front ends find a tile's entries:
//   rows  (grids of more than slr_splat_set_scan_max_tiles tiles; slr_splat_bin / prebinned calls): rowbin_kernel appends every
//         64-pixel row segment of the flow to the few tiles its footprints touch -- one returning 64-bit atomic per (segment, tile) =
//         list slot + the tile's exact entry count -- and its last workgroup writes the work plan: heavy tiles first, a tile of more
//         than SEG entries cut into ranges of its OUTPUT COLUMNS (a piece owns its pixels: no partial tiles, no combine);
//         op_rows_kernel walks exactly the listed rows (splat_rows.hpp).  3 launches: rows + plan, tile kernel, a normally empty
//         pass-by-pass launch for pieces that still hold more than SEG entries (the workspace arrives zeroed: see slr_splat_bin);
//   scan  (small grids): scan_box_kernel writes the destination box of every 8x64 block of source pixels; every output tile's
//         workgroup tests the boxes, lists the rows of the blocks that touch it in LDS and walks them with the same code.  Nothing
//         to zero, no plan.  A tile of more than SLR_SCAN_DEFER_AT entries -- a pile-up of a contracting flow -- writes its entries
//         out once and is rendered by the SINK launch (op_sink_kernel: tasks of one segment on many workgroups, slabs added up in
//         slot order; normally empty).  3 launches: boxes, tile kernel, sink launch.
#include "splat_rows.hpp"
#include "splat_ws.hpp"

#include <stdarg.h>
#include <atomic>

namespace slr {

static thread_local char g_err[512] = "";
#ifdef SLR_TRACE
long long *g_trace;
#endif
thread_local void *g_ev_start = nullptr, *g_ev_stop = nullptr;   // slr_splat_time_next (also armed for splat_clip.hip's launches)
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

using OpCfg = TileCfg<1, SLR_EPT_ONE, true, SLR_KREG_ROWS>;       // one flow: 1024 entries per workgroup, 6-byte records, 46 KiB of LDS
using OpPassCfg = TileCfg<1, SLR_EPT_DEFER, true, SLR_KREG_ROWS>; // the pass-by-pass launches: passes of 2048 entries, 86 KiB (most deferred pieces
                                                                  // are just over a segment and finish in one pass; these run on a near-empty chip)
static_assert(SLR_SCAN_DEFER_AT <= SLR_EPT_ONE * 512, "a tile rendered in place fits one segment");
using ScanCfg = TileCfg<1, SLR_EPT_ONE, true, SLR_KREG_ROWS, false, true>;     // the scan front end's kernels: + the balanced gather's sums (splat_tile.hpp: BalLane), 63 KiB (two workgroups per CU by their registers anyway)
constexpr int OP_SEG = OpCfg::SEG;
constexpr uint32_t OP_DEFER_WG = 64;               // workgroups of the pass-by-pass launch (x channel groups)
static_assert(4 * ROW_CAP * 4 + 2048 * 4 <= OpCfg::REC_BYTES, "row lists and the scan's candidate list live in the record area");

// =========================================================================== rows front end: binning + plan
// Work plan from the per-tile (entries, row segments) words; run by ONE workgroup of TILE_PIX work-items (the last one of
// rowbin_kernel) in ONE pass over the tiles.  A round covers 4 * TILE_PIX tiles: every work-item loads the words of 4
// CONSECUTIVE tiles together (one memory round trip per round), sums them locally, and two workgroup scans per round -- partial
// slots, then (heavy | other) items packed in one 64-bit word -- place them (a scan is a chain of cross-lane steps and
// barriers, ~0.5 us: one per tile and quantity made the plan 9.5 us of a 32 us kernel at 1920 tiles).
// Tiles in row-major order; on grids of more than one round of workgroups the heavy tiles (more than SLR_PLAN_HEAVY / 4 of an
// undisturbed tile's ~585 entries) go first: their items fill items[] from the front, everybody else's from the back
// (items[cap - 1 - k]), and the tile kernel reads item i < totals[5] from the front.  Segments of `seg` entries; a tile whose
// segments do not fit the partial-slot budget is left to one workgroup (nseg 0).
// (Round 4, measured and rejected: "tail fill" -- the item count rounded up to a multiple of the chip's 768 workgroup slots by cutting the
//  last whole tiles of the launch order into column halves, so that the last round of workgroups is full and short.  The traced lives say
//  the slots are 65 - 80 % busy (identity: 1920 lives of 34 us = 85 us of slot time in a 130 us kernel), but two halves cost 1.3 tiles
//  and the dispatcher refills slots one by one, not in rounds: call identity / t=30 / t=59 / incoherent 141-148 / 160-169 / 203-211 /
//  208-211 us without, 156 / 176 / 224 / 224 with.)
template <typename V>
__device__ __forceinline__ V exscan_tile_pix(V v, V *excl, V *wsum /*[TILE_PIX / 64]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    V inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const V o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    V woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TILE_PIX / 64; ++w) {
        const V q = wsum[w];
        if (w < wid) woff += q;
        total += q;
    }
    __syncthreads();
    *excl = woff + inc - v;
    return total;
}

__device__ __forceinline__ void rows_plan(const unsigned long long *__restrict__ rowcnt, unsigned long long *__restrict__ rowinfo, uint32_t nt, uint32_t seg, uint32_t heavy,
                                          uint32_t items_cap, ItemDesc *__restrict__ items, uint32_t *__restrict__ totals) {
    __shared__ unsigned long long wsum[TILE_PIX / 64];
    // The words were written by other workgroups' agent-scope atomics (performed at the memory side, before their arrival
    // atomics); this XCD's L2 may still hold the zeros of rows_zero_kernel.  They are read with agent-scope (sc1) loads, all four
    // of a round in flight together (as __hip_atomic_load the compiler waits after each).
    constexpr uint32_t PER = SLR_ROWS_PLAN_PER;
#define PSTAMP(k) do { } while (0)
    PSTAMP(0);
    const uint32_t heavy_thr = heavy ? (heavy * 585u) / 4u : 0xffffffffu;
    uint32_t run_heavy = 0, run_light = 0, run_extra = 0;
    const uint32_t extra_cap = items_cap - nt;                                // items beyond one per tile that items[] can hold
    for (uint32_t b = 0; b < nt; b += PER * TILE_PIX) {
        const uint32_t t0 = b + PER * threadIdx.x;
        unsigned long long w[PER], oh[PER];
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {                  // 8 agent-scope loads in flight, one wait
            const unsigned long long *pw_ = rowcnt + 2 * (size_t)(t0 + k < nt ? t0 + k : 0u);
            asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %2, off offset:8 sc1"
                         : "=&v"(w[k]), "=&v"(oh[k]) : "v"(pw_) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) asm volatile("" : "+v"(w[k]), "+v"(oh[k]));      // (read only after the wait)
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (t0 + k >= nt) w[k] = 0ull;
            // (the words are left ZERO for the workspace's next binning: a call does not have to zero them first)
            else {
                rowinfo[2 * (size_t)(t0 + k)] = w[k]; rowinfo[2 * (size_t)(t0 + k) + 1] = oh[k];      // (kept for slr_synth_group's two-flow plan)
                const_cast<unsigned long long *>(rowcnt)[2 * (size_t)(t0 + k)] = 0ull; const_cast<unsigned long long *>(rowcnt)[2 * (size_t)(t0 + k) + 1] = 0ull;
            }
        }
        PSTAMP(1);
        unsigned long long mine = 0;                                          // (heavy items << 32) | other items of my 4 tiles
        uint32_t ns[PER], io[PER], xo[PER];
        unsigned long long pcs[PER];                                          // pieces of the tile: (first octant | octants << 4), 8 bits each
        bool hv[PER];
        unsigned long long extra = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t cnt = (uint32_t)(w[k] >> 32);
            // Pieces of a heavy tile = ranges of its 8 column octants (8 output columns each), cut greedily so that no piece's octant
            // counts add up to more than 7/8 of a segment (an entry on an octant boundary counts in both: the sum bounds the piece
            // from above).  Columns, not rows: a footprint is two pixels wide and two high, so 8 pieces by rows stage 1.78x the
            // tile's entries, by columns 1.10x.  An octant that holds more than a segment by itself makes a piece that its workgroup
            // finds too long and hands to the pass-by-pass launch.
            pcs[k] = 0x80ull;                                                 // one piece: octants [0, 8)
            ns[k] = t0 + k >= nt ? 0u : 1u;
            if (ns[k] && cnt > seg) {
                const uint32_t limit = (seg * (uint32_t)SLR_ROWS_FILL) / 8u;
                // (the histogram is an estimate -- units of 16 entries, small appends left out -- scaled to the tile's count + 1/8
                // for the entries that sit on an octant boundary and count twice)
                uint32_t hsum = 0;
#pragma unroll
                for (uint32_t o = 0; o < 8; ++o) hsum += (uint32_t)(oh[k] >> (8 * o)) & 0xffu;
                const uint32_t osum = cnt + cnt / 8u;
                // (8 bits per octant: an octant byte that passed 255 units -- more than ~4000 entries -- has carried into its neighbour
                // and the histogram no longer adds up to the tile's exact count: cut such a tile into its 8 octants, the pieces that
                // are still too long go pass by pass)
                const bool hist_ok = hsum * 32u >= cnt;
                // (float arithmetic: these are estimates, and 64-bit integer divisions cost the one planning workgroup 2.7 us)
                const float scale = (float)osum / (float)(hsum ? hsum : 1u);
                const uint32_t even = (uint32_t)((float)osum / ceilf((float)osum / (float)limit));   // pieces of about equal weight, not one full + a rest
                uint32_t start = 0, sum = 0, np = 0;
                unsigned long long p = 0;
#pragma unroll
                for (uint32_t o = 0; o < 8; ++o) {
                    const uint32_t co = (uint32_t)((float)((uint32_t)(oh[k] >> (8 * o)) & 0xffu) * scale);
                    if ((sum + co > limit || sum + co / 2u >= even) && o > start) { p |= (unsigned long long)(start | ((o - start) << 4)) << (8 * np); ++np; start = o; sum = 0; }
                    sum += co;
                }
                p |= (unsigned long long)(start | ((8u - start) << 4)) << (8 * np); ++np;
                if (!hist_ok) { p = 0x1716151413121110ull; np = 8; }
                pcs[k] = p; ns[k] = np;
            }
            xo[k] = (uint32_t)extra;
            extra += ns[k] ? ns[k] - 1u : 0u;
        }
        unsigned long long xex;
        const uint32_t xtot = (uint32_t)exscan_tile_pix<unsigned long long>(extra, &xex, wsum);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            const uint32_t cnt = (uint32_t)(w[k] >> 32);
            if (ns[k] > 1u && run_extra + (uint32_t)xex + xo[k] + ns[k] - 1u > extra_cap) { ns[k] = 1u; pcs[k] = 0x80ull; }   // items[] is full: one piece (pass by pass)
            hv[k] = ns[k] && cnt > heavy_thr;
            io[k] = hv[k] ? (uint32_t)(mine >> 32) : (uint32_t)mine;
            mine += hv[k] ? (unsigned long long)ns[k] << 32 : (unsigned long long)ns[k];
        }
        unsigned long long iex;
        const unsigned long long itot = exscan_tile_pix<unsigned long long>(mine, &iex, wsum);
        PSTAMP(3);
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            if (!ns[k]) continue;
            ItemDesc d;
            d.tile = t0 + k; d.cnt0 = (uint32_t)(w[k] >> 32); d.cnt1 = (uint32_t)w[k]; d.off0 = 0; d.off1 = 0; d.partoff = 0;
            const uint32_t at = hv[k] ? run_heavy + (uint32_t)(iex >> 32) + io[k] : run_light + (uint32_t)iex + io[k];
            for (uint32_t q = 0; q < ns[k]; ++q) {
                const uint32_t pc = (uint32_t)(pcs[k] >> (8 * q)) & 0xffu;
                d.seg = pc & 0xfu;                                            // first column octant of the piece
                d.nseg = pc >> 4;                                             // its octants (8 = the whole tile)
                items[hv[k] ? at + q : items_cap - 1u - (at + q)] = d;
            }
        }
        run_heavy += (uint32_t)(itot >> 32);
        run_light += (uint32_t)itot;
        run_extra += xtot;
        PSTAMP(4);
    }
    // totals[4]: pieces that turn out to need more than one pass (appended by their workgroups, read by the WHOLE launch)
    if (threadIdx.x == 0) { totals[0] = run_heavy + run_light; totals[1] = 0; totals[3] = 0; totals[4] = 0; totals[5] = run_heavy; totals[6] = 0; }
}

// grid: N * tiles_x * ceil(tiles_y / ROWBIN_R) workgroups; a workgroup covers ROWBIN_R vertically adjacent source tiles (wave w: rows
// w, w + 8, ... of the block, all their flow loads in flight together, their appends in ONE atomic instruction): a wave's life is
// one load round trip + one atomic round trip however many rows it carries, and 1920 one-tile workgroups took 2.5 rounds of that.
constexpr int ROWBIN_R = SLR_ROWBIN_R;
__global__ __launch_bounds__(TILE_PIX) void rowbin_kernel(const float *__restrict__ flow, unsigned long long *__restrict__ rowcnt, unsigned long long *__restrict__ rowinfo,
                                                          RowRec *__restrict__ rowlist, int H, int W, int tiles_x, int tiles_y,
                                                          uint32_t nt, uint32_t *__restrict__ ctl, uint32_t *__restrict__ arrive1,
                                                          uint32_t seg, uint32_t heavy, uint32_t items_cap,
                                                          ItemDesc *__restrict__ items, uint32_t *__restrict__ totals) {
    const int tiles = tiles_x * tiles_y, by_n = (tiles_y + ROWBIN_R - 1) / ROWBIN_R, per_n = tiles_x * by_n;
    const int b = blockIdx.x, n = b / per_n, bl = b - n * per_n;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stx = bl % tiles_x, y_base = (bl / tiles_x) * ROWBIN_R * TILE_H + wid, x = stx * TILE_W + lane;
    const float *fl = flow + (size_t)n * 2 * H * W;
    float fx[ROWBIN_R], fy[ROWBIN_R];
#pragma unroll
    for (int r = 0; r < ROWBIN_R; ++r) {
        const int y = y_base + r * TILE_H;
        const size_t q = (y < H && x < W) ? (size_t)y * W + x : 0;
        fx[r] = fl[q];
        fy[r] = fl[(size_t)H * W + q];
    }
    // distinct tiles of a row's 64 footprints, one per round: the first lane with something left names a tile, a ballot counts the
    // lanes that touch it; round k's (tile, row, hits) is parked in lane k and all appends go out as ONE atomic instruction
    unsigned long long *cnt_n = rowcnt + 2 * (size_t)n * tiles;
    RowRec *list_n = rowlist + (size_t)n * tiles * ROW_CAP;
    int my_tile = -1, my_y = 0;
    uint32_t my_cnt = 0;
    unsigned long long my_hist = 0;                      // hits per column octant of the tile in units of 16, 8 bits each
    int k = 0;
    auto flush = [&]() {
        if (my_tile >= 0) {
            const unsigned long long old = atomicAdd(cnt_n + 2 * (size_t)my_tile, 1ull | ((unsigned long long)my_cnt << 32));
            if (my_hist) atomicAdd(cnt_n + 2 * (size_t)my_tile + 1, my_hist);       // (no return value: fire and forget)
            const uint32_t slot = (uint32_t)old;
            if (slot < (uint32_t)ROW_CAP) list_n[(size_t)my_tile * ROW_CAP + slot] = RowRec{(uint32_t)my_y, ((uint32_t)stx << 8) | my_cnt};
        }
        my_tile = -1;
        k = 0;
    };
#pragma unroll
    for (int r = 0; r < ROWBIN_R; ++r) {
        const int y = y_base + r * TILE_H;
        if (y >= H) break;                                   // (wave-uniform)
        int t0 = -1, t1 = -1, t2 = -1, t3 = -1;             // the <= 4 tiles this pixel's footprint touches
        uint32_t cm_a = 0, cm_b = 0;                        // column octants (bits) it touches in the left / right of them
        if (x < W) {
            // tile column a holds corner column x0 (if in the image) and x0 + 1 when it lies in the same tile column; tile column b
            // (valid only when distinct) holds x0 + 1.  Octant = 8 output columns (the finest piece of a heavy tile).
            const BinFoot f = bin_footprint(fx[r], fy[r], x, y, H, W, tiles_x);
            t0 = f.t0; t1 = f.t1; t2 = f.t2; t3 = f.t3; cm_a = f.cm_a; cm_b = f.cm_b;
        }
        for (;;) {
            const int cand = (int)min(min((uint32_t)t0, (uint32_t)t1), min((uint32_t)t2, (uint32_t)t3));      // (any pending tile will do; -1 = none)
            const unsigned long long pend = __ballot(cand >= 0);
            if (!pend) break;
            const int leader = __ffsll((long long)pend) - 1;
            const int T = __builtin_amdgcn_readlane(cand, leader);
            const bool e0 = t0 == T, e1 = t1 == T, e2 = t2 == T, e3 = t3 == T;
            const unsigned long long hmask = __ballot(e0) | __ballot(e1) | __ballot(e2) | __ballot(e3);        // (scalar ORs of the four compare masks)
            const uint32_t c = (uint32_t)__popcll(hmask);
            const uint32_t lm = ((e0 | e2) ? cm_a : 0u) | ((e1 | e3) ? cm_b : 0u);   // column octants of T this lane touches
            // Column-octant histogram of the tile (what the plan cuts heavy tiles by): ONE more atomic per append, 8 bits per octant
            // in units of 16 entries with a pseudo-random rounding offset (unbiased: a tile's sum over its ~50 appends is what
            // matters; two 16-bit-per-octant words cost +7 us per call at 46 k appends).  Appends of fewer than 8 hits -- the
            // one-column overlaps into the neighbouring tile, half of all appends -- stay out of it.
            // (This loop is what the kernel's time grows with -- a bent row touches up to ~8 tiles, 12 us per workgroup at Euler
            // t=59 against 6 on the identity flow -- so the small appends take a short way: the union of <= 7 lanes' masks.)
            uint32_t rm = 0;
            unsigned long long hist = 0;
            if (c >= 8u) {                                   // (wave-uniform)
                const uint32_t rnd = ((uint32_t)y * 2654435761u + (uint32_t)T * 40503u) >> 16;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const uint32_t co = (uint32_t)__popcll(__ballot((lm >> o) & 1u));
                    rm |= co ? 1u << o : 0u;
                    hist |= (unsigned long long)((co + ((rnd >> o) & 15u)) >> 4) << (8 * o);
                }
            } else {
                for (unsigned long long m = hmask; m; m &= m - 1ull)
                    rm |= (uint32_t)__builtin_amdgcn_readlane((int)lm, __ffsll((long long)m) - 1);
            }
            t0 = e0 ? -1 : t0;
            t1 = e1 ? -1 : t1;
            t2 = e2 ? -1 : t2;
            t3 = e3 ? -1 : t3;
            {                                                // round k's append is parked in lane k (wave-uniform values: v_writelane, splat_types.hpp)
                int v_cnt = (int)my_cnt, v_lo = (int)(uint32_t)my_hist, v_hi = (int)(uint32_t)(my_hist >> 32);
                write_lane5(k, my_tile, T, v_cnt, (int)c, my_y, y | (int)(rm << 24), v_lo, (int)(uint32_t)hist, v_hi, (int)(uint32_t)(hist >> 32));
                my_cnt = (uint32_t)v_cnt; my_hist = (unsigned long long)(uint32_t)v_lo | ((unsigned long long)(uint32_t)v_hi << 32);
            }
            if (++k == 64) flush();
        }
    }
    flush();
    // ---- the last workgroup to get here plans the call (every append above has returned: its value was used)
    // (two levels: thousands of returning atomics on ONE word are served one after the other -- 35 us at 1920 workgroups)
    __shared__ uint32_t last;
    __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0): the fire-and-forget histogram adds have been performed at the memory side too
    __syncthreads();
    if (tid == 0) {
        const uint32_t grp = blockIdx.x >> 6, ngrp = (gridDim.x + 63u) >> 6;
        const uint32_t members = min(64u, gridDim.x - (grp << 6));
        uint32_t l = 0;
        if (atomicAdd(&arrive1[(size_t)grp * 32u], 1u) == members - 1u) l = atomicAdd(&ctl[0], 1u) == ngrp - 1u ? 1u : 0u;
        last = l;
    }
    __syncthreads();
    if (!last) return;
    rows_plan(rowcnt, rowinfo, nt, seg, heavy, items_cap, items, totals);
    // everybody has arrived: the arrival counters go back to zero for the workspace's next binning
    for (uint32_t i = tid; i < ((gridDim.x + 63u) >> 6); i += TILE_PIX) arrive1[(size_t)i * 32u] = 0u;
    if (tid == 0) ctl[0] = 0u;
}

// The counters rowbin_kernel adds to (rowcnt words, arrival counters).  rowbin_kernel's planning workgroup leaves them zero again, so a
// workspace that was zeroed once (slr_splat_workspace_init) and is only used through this library never needs this kernel: calls that
// say so (SLR_WS_CLEAN) skip it; for any other workspace -- the caller's memory, nothing can be assumed about it -- it runs first.
__global__ __launch_bounds__(256) void rows_zero_kernel(unsigned long long *__restrict__ rowcnt, uint32_t nt, uint32_t *__restrict__ ctl,
                                                        uint32_t *__restrict__ arrive1) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nt) { rowcnt[2 * (size_t)i] = 0ull; rowcnt[2 * (size_t)i + 1] = 0ull; }
    if (i < (nt + 63u) / 64u) arrive1[(size_t)i * 32u] = 0u;    // first-level arrival counters of rowbin_kernel: one per 64
                                                                 // workgroups, each on its own 128-byte line (atomics on one line are served
                                                                 // one after the other: 1920 arrivals on one line cost 14 us)
    if (i < 16u) ctl[i] = 0u;
}


// Everything in front of the row lists (counters, plans, boxes) of a fresh workspace: zero.
__global__ __launch_bounds__(256) void ws_zero_kernel(unsigned long long *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// =========================================================================== scan front end: destination boxes
// Small grids spend their time in the latency chains of dependent launches; the scan front end needs ONE small kernel before the tile
// kernel: every 8x64 block of SOURCE pixels ("source tile") gets the bounding box of the NW corners its pixels splat to (no atomics,
// nothing to zero, 3 us).  An output tile's workgroup then tests all boxes (16 bytes each, L2-resident) and walks the rows of the few
// source tiles whose box touches it.  Any flow is handled exactly: a box that covers everything just means more candidates.
__global__ __launch_bounds__(TILE_PIX) void scan_box_kernel(const float *__restrict__ flow, SrcBox *__restrict__ box, int H, int W,
                                                            int tiles_x, int tiles, uint32_t *__restrict__ totals) {
    const int t = blockIdx.x, n = t / tiles, tl = t - n * tiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (t == 0 && tid < 8) { totals[tid] = 0u; totals[16 + tid] = 0u; }   // the deferred list of the tile kernel that follows, the sink launch's task counters (a kernel boundary makes the zeros visible)
    const int y = (tl / tiles_x) * TILE_H + wid, x = (tl % tiles_x) * TILE_W + lane;
    int bx0 = 0x7fffffff, bx1 = -0x7fffffff, by0 = 0x7fffffff, by1 = -0x7fffffff;
    if (y < H && x < W) {
        const float *f = flow + (size_t)n * 2 * H * W + (size_t)y * W + x;
        const Corners c = make_corners(f[0], f[(size_t)H * W], x, y);
        // some corner of the footprint lies inside the image (the same test as footprint_tiles)
        if (c.ok && c.x0 >= -1 && c.x0 <= W - 1 && c.y0 >= -1 && c.y0 <= H - 1) { bx0 = bx1 = c.x0; by0 = by1 = c.y0; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, d)); bx1 = max(bx1, __shfl_xor(bx1, d));
        by0 = min(by0, __shfl_xor(by0, d)); by1 = max(by1, __shfl_xor(by1, d));
    }
    __shared__ int red[TILE_H][4];
    if (lane == 0) { red[wid][0] = bx0; red[wid][1] = bx1; red[wid][2] = by0; red[wid][3] = by1; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < TILE_H; ++w) {
            bx0 = min(bx0, red[w][0]); bx1 = max(bx1, red[w][1]); by0 = min(by0, red[w][2]); by1 = max(by1, red[w][3]);
        }
        SrcBox b; b.x0 = bx0; b.x1 = bx1; b.y0 = by0; b.y1 = by1;
        box[t] = b;
    }
}

// =========================================================================== tile kernels

// planes [cb, ce) of this workgroup: on grids smaller than the chip (gridDim.y > 1) a tile's planes are dealt to 2-4 workgroups, each
// builds the tile's records itself (phase 1 is the cheap part) -- a tile is one workgroup whose chunk pipeline nothing overlaps with
// (256x480, C = 64: 37.5 -> 33.5 us with 2 groups; 128x240: 34 -> 21 us with 4).  Groups start on a multiple of 8 planes.
__device__ __forceinline__ bool channel_group(int C, int &cb, int &ce) {
    const int cper = (((C + (int)gridDim.y - 1) / (int)gridDim.y + 7) / 8) * 8;
    cb = (int)blockIdx.y * cper;
    ce = min(C, cb + cper);
    return cb < C;
}
// The sink launch's rule (8 groups): the planes in units of 8, dealt as evenly as whole units go; the LAST groups take the odd units, so
// that the partial unit at the end (C = 65: one plane) rides along with a full one -- 65 planes are 7 x 8 + 9.  (The rule above makes
// them 4 x 16 + 1 and three groups without planes: a task streamed 16 planes where 8 - 9 do; training shape t=59, sink launch 58.7 ->
// 51.4 us.  In the tile kernels the even split measured SLOWER -- C = 65 over 2 groups, 32 + 33 against 40 + 25 planes: 54.6 against
// 50.8 us at the training shape -- they keep the rule above.)
__device__ __forceinline__ bool channel_range(int C, int G, int g, int &cb, int &ce) {
    const int U = (C + 7) >> 3, base = U / G, first_big = G - (U - base * G);
    const int ub = g * base + max(0, g - first_big), un = base + (g >= first_big ? 1 : 0);
    cb = ub * 8;
    ce = min(C, cb + un * 8);
    return cb < ce;
}

struct OpArgs {
    TileShared s; TileFrame f;
    uint32_t *sink_cnt;            // scan front end: [items_cap][16] per deferred piece: arrivals of the sink launch per channel group [0..7], its tasks [8]
    float *sink_pool;              // ... slabs: [sink_cap][rows][TILE_PIX] handed out to the deferred pieces, then the emergency slabs [SINK_P][rows][TILE_PIX]
    uint32_t sink_cap, sink_t;     // ... slabs of the pool; most task slots a piece gets (rows = planes + channel groups of the sink launch)
    float4 *sink_ent;              // ... [sink_ent_cap] the entries of the deferred pieces
    uint32_t sink_ent_cap;
    uint32_t sink_x, sink_groups;  // ... the sink launch's piece slots and channel groups (its grid: sink_x * sink_groups * SINK_T workgroups, one dimension)
};

// rows front end.  grid.x: a multiple of 8 * SLR_XCD_GROUP blocks covering the plan's items (surplus workgroups exit at once).
// PASSES = false: one piece per workgroup, no loop over work (80 VGPRs: three workgroups per CU); a piece of more than SEG entries is
// appended to the deferred list.  PASSES = true: OP_DEFER_WG workgroups walk the deferred list pass by pass (normally it is empty).
template <bool NORM, bool MAXOP, bool PASSES>
__global__ __launch_bounds__(TT, PASSES ? 1 : SLR_WAVES_ROWS) void op_rows_kernel(OpArgs a) {
    using Cfg = std::conditional_t<PASSES, OpPassCfg, OpCfg>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const TileLds<Cfg> L(smem);
    const TileShared &s = a.s;
    const TileFrame &f = a.f;
    const int tid = threadIdx.x;
#ifdef SLR_TRACE
    const long long t_entry = (long long)wall_clock64();
#endif
    if (PASSES && f.totals[4] == 0u) return;               // the normal case: an empty launch whose workgroups do one scalar load (nothing to reset)
    int cb, ce;
    const bool has_planes = channel_group(s.C, cb, ce);
    if (!PASSES && !has_planes) return;                    // (PASSES: a workgroup without planes still ARRIVES below -- the reset counts the whole grid)
    const TileScalars k = tile_scalars(s, f);
    if (!PASSES) {
        // One workgroup per item, no loop over work.  (Round 4, traced with the constant clock: a slot stays empty for ~4.5 us between the end
        // of one workgroup and the first instruction of the next, and the new one needs 2.5 us to know its item -- 630-680 of the 768 slots
        // are occupied in the steady state.  As many workgroups as the chip holds, each walking items blockIdx.x, + grid.x, ... was measured
        // and rejected: the loop's live kernel arguments cost 19 spilled registers and the static shares lose the dispatcher's balancing --
        // call identity / t=30 / t=59 / incoherent 150 / 183 / 224 / 223 us against 146-153 / 171-180 / 216-224 / 214-219.)
        const uint32_t item = xcd_item(blockIdx.x);
        if (item >= f.totals[0]) return;
        const uint32_t nh = f.totals[5];                   // heavy items sit at the front of items[], the rest at its back
        const uint32_t at = item < nh ? item : f.items_cap - 1u - (item - nh);
        const Piece p = make_piece<Cfg>(s, f.items[at]);
        T_NOTE(s, 56, wall_clock64());                     // (the constant 100 MHz clock: comparable across XCDs, unlike the shader clock of T_STAMP)
#ifdef SLR_TRACE
        T_NOTE(s, 58, t_entry);
#endif
        if (!rows_piece_once<Cfg, false, NORM, MAXOP, false>(s, f, L, p, tid, k, cb, ce) && tid == 0 && blockIdx.y == 0)
            f.defer[atomicAdd(f.totals + 4, 1u)] = at;     // (one entry per piece: every channel group gets here)
        T_NOTE(s, 57, wall_clock64());
    } else {
        const uint32_t ndef = has_planes ? f.totals[4] : 0u;
        for (uint32_t q = blockIdx.x; q < ndef; q += gridDim.x)
            rows_piece_passes<Cfg, false, NORM, MAXOP, false>(s, f, L, make_piece<Cfg>(s, f.items[f.defer[q]]), tid, k, cb, ce);
        // the last workgroup to get here empties the deferred list for the plan's next use (prebinned calls share one plan).  EVERY
        // workgroup of the grid arrives -- also those whose channel group is empty (C = 36, 65, 72, 80 with 8 groups): counting only some
        // of them left the list in place for the next prebinned call, which then walked the same pieces again
        __syncthreads();
        if (tid == 0 && atomicAdd(f.totals + 6, 1u) == gridDim.x * gridDim.y - 1u) { f.totals[4] = 0u; f.totals[6] = 0u; }
    }
}

// scan front end: one workgroup per output tile (x channel groups).  The boxes of all source tiles are tested 2048 at a time, the
// rows of the candidates are listed in LDS 32 candidates at a time (wave w = row w of each: coalesced 256-byte loads) and walked by
// rows_walk.  Optimistic first: one LDS atomic per wave and row hands out the entry slots; a tile that turns out to hold more than SEG
// entries is walked again in passes of SEG entries with reproducible ordinals (a count walk, then one emitting walk per pass).
// The candidate source tiles of a piece among source tiles [base, base + 2048): their boxes tested against the piece's columns, the hits
// as an ordered list in LDS (clist, in the record area).  Returns their number; ends with a barrier.
template <class Cfg>
__device__ __forceinline__ uint32_t scan_candidates(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid, int base) {
    uint32_t *cmask = reinterpret_cast<uint32_t *>(L.off);         // [64] candidate bits of a block of 2048 source tiles (off[] is free until phase 1b)
    uint32_t *clist = L.rl + 4 * ROW_CAP;                           // [2048] candidate source tiles of the block, in index order
    const SrcBox *boxes = f.box + (size_t)p.n * s.tiles;
    SrcBox bx4[2048 / TT];
#pragma unroll
    for (int q = 0; q < 2048 / TT; ++q) {                           // the box loads do not depend on LDS: issue them first
        const int st = base + tid + q * TT;
        bx4[q] = boxes[st < s.tiles ? st : 0];
    }
    if (tid < 64) cmask[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2048 / TT; ++q) {
        const int st = base + tid + q * TT;
        const SrcBox b = bx4[q];
        // (against the piece's own columns: where a flow contracts -- the pile-ups that get cut into pieces -- source boxes are narrow)
        if (st < s.tiles && b.x1 >= p.tx0 + p.pca - 1 && b.x0 <= p.tx0 + p.pcb - 1 && b.y1 >= p.ty0 - 1 && b.y0 <= p.ty0 + TILE_H - 1)
            atomicOr(&cmask[(st - base) >> 5], 1u << (st & 31));
    }
    __syncthreads();
    // the bit mask -> the ordered candidate list (lane l of wave 0 owns word l; a wave scan places its bits)
    const int lane = tid & 63;
    const uint32_t word = cmask[lane];
    const uint32_t pc = (uint32_t)__popc(word);
    const uint32_t inc = wave_incl_scan(pc, lane);
    const uint32_t nc = (uint32_t)__shfl(inc, 63);
    if (tid < 64) {
        uint32_t w = word, at = inc - pc;
        while (w) {
            const int bit = __ffs((int)w) - 1;
            w &= w - 1;
            clist[at++] = (uint32_t)(base + lane * 32 + bit);
        }
    }
    __syncthreads();
    return nc;
}

// rows [r0, r0 + n) of the flattened (candidate, row) space -> the row-segment list in LDS (index 8 * candidate + row: wave w walks row w of each)
__device__ __forceinline__ void scan_row_words(const TileShared &s, uint32_t st, int row, uint32_t &w0, uint32_t &w1) {
    const uint32_t sty = st / (uint32_t)s.tiles_x, stx = st - sty * (uint32_t)s.tiles_x;
    const uint32_t sy = sty * TILE_H + (uint32_t)row;
    w0 = sy < (uint32_t)s.H ? sy | 0xff000000u : 0u;               // (no octants: a row past the image is skipped)
    w1 = stx << ROWW_STX;
}

// scan front end: one workgroup per output tile (x channel groups).  The boxes of all source tiles are tested 2048 at a time, the
// rows of the candidates are listed in LDS 32 candidates at a time (wave w = row w of each: coalesced 256-byte loads) and walked by
// rows_walk: one LDS atomic per wave and row hands out the entry slots (the total is exact even when it exceeds SEG).
// MODE 1: one LDS atomic per wave and row hands out the entry slots (the total, L.misc[0], is exact even when it exceeds SEG; entries past
// SEG are dropped); MODE 2 + GLB: the second walk of a piece that turned out to be a sink -- reproducible ordinals (wave_base from the
// first walk's per-wave hits), every entry written to `gent`.  Returns this wave's hits; ntask: the candidate-pair tasks of the piece.
template <class Cfg, int MODE, bool GLB>
__device__ __forceinline__ uint32_t scan_collect(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, Piece &p, int tid,
                                                 uint32_t wave_base, float4 *gent, uint32_t &ntask) {
    const uint32_t *clist = L.rl + 4 * ROW_CAP;
    uint32_t wcount = 0;
    ntask = 0;
    for (int base = 0; base < s.tiles; base += 2048) {
        const uint32_t nc = scan_candidates<Cfg>(s, f, L, p, tid, base);
        ntask += (nc + 1u) / 2u;
        if (base == 0 && MODE == 1) { T_STAMP(s, 1); T_NOTE(s, 61, nc); }
        for (uint32_t c0 = 0; c0 < nc; c0 += ROW_CAP / TILE_H) {
            const uint32_t n = min(nc - c0, (uint32_t)(ROW_CAP / TILE_H));
            if ((uint32_t)tid < n * TILE_H) scan_row_words(s, clist[c0 + (uint32_t)tid / TILE_H], tid % TILE_H, L.rl[tid], L.rl[ROW_CAP + tid]);
            __syncthreads();
            p.len0 = p.n0 = n * TILE_H;
            wcount += rows_walk<Cfg, MODE, true, false, GLB>(s, f, L, p, tid, wave_base + wcount, 0u, GLB ? 0xffffffffu : (uint32_t)Cfg::SEG, gent);
            __syncthreads();
        }
    }
    return wcount;
}

// One workgroup per output tile, or per column piece of it (grid.z), x channel groups.  A piece that turns out to hold more than SEG
// entries -- a pile-up of a contracting flow -- is not rendered here: it is appended to the deferred list, and the SINK launch that
// follows (normally empty) renders it with many workgroups at once.
// (Measured and rejected in rounds 4 / 5: the deferred pieces as TAIL blocks of the same launch behind an arrival counter -- config C2
//  37 -> 49.5 us, 512 arrival atomics on one word ~ 20 us; column pieces in the first launch: every piece repeats the candidate walk.)
template <bool NORM, bool MAXOP>
__global__ __launch_bounds__(TT, SLR_WAVES_SCAN) void op_scan_kernel(OpArgs a) {
    using Cfg = ScanCfg;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const TileLds<Cfg> L(smem);
    const TileShared &s = a.s;
    const TileFrame &f = a.f;
    const int tid = threadIdx.x;
    int cb, ce;
    if (!channel_group(s.C, cb, ce)) return;
    const TileScalars k = tile_scalars(s, f);
    // grid.z column pieces per tile (1, 2, 4 or 8: slr_splat_set_scan_shape; 1 by default -- every piece repeats the candidate walk)
    ItemDesc it = {};
    it.tile = xcd_item(blockIdx.x); it.nseg = 8u / gridDim.z; it.seg = blockIdx.z * it.nseg;
    if (it.tile >= (uint32_t)s.N * (uint32_t)s.tiles) return;
    Piece p = make_piece<Cfg>(s, it);
    T_STAMP(s, 0);
    L.cnt[tid] = 0;
    if (tid == 0) L.misc[0] = 0;
    __syncthreads();
    uint32_t ntask;
    const uint32_t whits = scan_collect<Cfg, 1, false>(s, f, L, p, tid, 0u, nullptr, ntask);
    const uint32_t total = L.misc[0];
    T_STAMP(s, 2);
    T_NOTE(s, 60, total);
    if (total <= (uint32_t)SLR_SCAN_DEFER_AT) {
        const rsrc_t rin = sample_planes(s, p, k.hw4);
        PixelSums sums = {0.0f, 0.0f, 0.0f};
        EntryRegs<Cfg> e;
        float preA[Cfg::EPT][4], preB[Cfg::EPT][4];
        build_records<Cfg, NORM, false>(s, L, p, tid, total, rin, k.hw4, cb, ce - 1, k.shift, k.sc0, k.sc1, e, preA, preB);
        T_STAMP(s, 6);
        stream_planes<Cfg, NORM, MAXOP, false, false, false, true>(s, f, L, p, tid, rin, k.hw4, cb, ce, e, preA, preB, sums, true, true);
        T_STAMP(s, 59);
        return;
    }
    // a sink.  Its first channel group appends it to the deferred list (tile | first octant << 20 | log2 octants << 23; arrival words at
    // zero), takes room for its entries in the call's entry array and walks the candidates once more, writing EVERY entry out at its
    // reproducible ordinal: the sink launch then cuts the piece into tasks of exactly SEG entries without walking anything.  No room left
    // (a flow with hundreds of sinks): the sink launch cuts it by candidate pairs instead.
    if (blockIdx.y != 0) return;
    uint32_t wb;
    const uint32_t all = wave_bases<Cfg>(L, tid, whits, wb);      // (== total)
    if (tid == 0) {
        const uint32_t q = atomicAdd(f.totals + 4, 1u), off = atomicAdd(f.totals + 7, all);
        const bool room = off <= a.sink_ent_cap && all <= a.sink_ent_cap - off;
        f.defer[q] = it.tile | (it.seg << 20) | ((uint32_t)(31 - __clz((int)it.nseg)) << 23);
        uint32_t *hd = a.sink_cnt + (size_t)q * 16u;
#pragma unroll
        for (int i = 0; i < 8; ++i) hd[i] = 0u;
        hd[8] = ntask; hd[9] = all; hd[10] = room ? off : 0xffffffffu;
        // its task slots: as many slabs of the pool as it has tasks (at most sink_t), or none (one workgroup per channel group, emergency slab)
        const uint32_t tasks = room ? (all + (uint32_t)Cfg::SEG - 1u) / (uint32_t)Cfg::SEG : ntask;
        const uint32_t want = min(a.sink_t, max(tasks, 1u));
        const uint32_t base = want ? atomicAdd(f.totals + 6, want) : a.sink_cap;
        hd[11] = (want && base <= a.sink_cap && want <= a.sink_cap - base) ? base : 0xffffffffu;
        L.misc[10] = room ? off : 0xffffffffu;
    }
    __syncthreads();
    const uint32_t off = L.misc[10];
    if (off != 0xffffffffu) scan_collect<Cfg, 2, true>(s, f, L, p, tid, wb, a.sink_ent + off, ntask);
    T_STAMP(s, 55);
}

// The SINK launch (grid: SLR_SINK_GRID workgroups that take their (piece, task slot, channel group) from an ordered task list -- below; normally the
// deferred list is empty and every workgroup ends after one scalar load).  A deferred piece holds thousands of entries -- Euler-integrated flows pile hundreds of source pixels onto a few
// output pixels; on a 256 x 256 training crop at t = 59 two tiles receive 12 000 entries each and one pixel 4 000 -- and walking them pass by
// pass in ONE workgroup (rounds 1-5) left the chip idle behind it: 116 us for a C2-sized call whose other tiles take 29, 670 us at the
// training shape.  Here the piece is cut by SOURCE: task k = candidate source tiles 2k, 2k + 1 of the piece's ordered candidate list
// (16 row segments: at most 1024 entries, one segment of LDS, no count pass), tasks dealt round-robin to the piece's task slots; a
// slot's workgroup renders its tasks one after the other into a SLAB of its own (un-normalised sums of its channel group's planes + the
// normaliser, accumulated through its own earlier stores: stream_planes<SLAB>), and the last workgroup of a (piece, channel group) to
// arrive adds the slabs up in slot order -- reproducible, no float atomics -- normalises and writes the piece's pixels.
// (First built with agent-scope fp32 atomicAdds into the output instead of slabs: correct, and 375 us for the C2-sized smooth case --
//  8 M memory-side atomics at ~22 G/s.)
// Slabs: a deferred piece takes as many slabs of the pool as it has tasks (at most sink_t; one atomic when it is deferred); a piece that finds the
// pool empty is rendered by ONE workgroup per channel
// group (task slot 0) out of that workgroup's emergency slab -- correct for any flow, slow only for flows with hundreds of sinks.
constexpr uint32_t SINK_P = SLR_SINK_PIECES, SINK_T = SLR_SINK_TASKS, SINK_MINE = 2048 / 2;
#ifdef SLR_TRACE      // per-workgroup wall-clock stamps of the sink launch (tools/dev/trace_sink.py): 16 words per workgroup behind the tile kernels' area
#define SINK_STAMP(slot, v) do { if (s.trace && threadIdx.x == 0) s.trace[(size_t)16384 * 64 + (size_t)blockIdx.x * 16 + (slot)] = (long long)(v); } while (0)
#else
#define SINK_STAMP(slot, v) do { } while (0)
#endif
static_assert(SINK_T <= 16, "slab bits of the arrival word");
template <bool NORM, bool MAXOP>
__global__ __launch_bounds__(TT, 4) void op_sink_kernel(OpArgs a) {
    using Cfg = ScanCfg;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    __shared__ uint32_t mine[2 * SINK_MINE];              // this workgroup's tasks of a candidate block (build_records overwrites the list)
    __shared__ uint32_t arrived, next_j;
    const TileLds<Cfg> L(smem);
    const TileShared &s = a.s;
    const TileFrame &f = a.f;
    const int tid = threadIdx.x;
    const uint32_t ndef = f.totals[4];
    if (ndef == 0u) return;                                // the normal case: an empty launch whose workgroups do one scalar load
    SINK_STAMP(0, wall_clock64());
    const TileScalars k = tile_scalars(s, f);
    const uint32_t *clist = L.rl + 4 * ROW_CAP;
    const size_t hw = (size_t)s.H * s.W;
    const uint32_t lin = blockIdx.x;
    // Which (piece, task slot, channel group) a workgroup renders.  The tasks that exist come FIRST in dispatch order: a slot of the
    // chip changes hands in microseconds, and behind a grid of pieces x groups x SINK_T slots of which a few hundred have work (most
    // pieces are one task) the later slots' workgroups waited for thousands of empty ones to pass (traced: half of the working
    // workgroups of a C2-sized call started 5 - 19 us late, 35 us at the training shape).  Rounds of R pieces of the deferred list; in
    // a round every workgroup derives the same ordered list of (slot z, piece) pairs whose piece has more than z task slots from the
    // pieces' headers (one lane each, SINK_T ballots) and takes the entry its index names.  With 8 channel groups the list is kept
    // per XCD (workgroup lin runs on XCD lin % 8): entry j = lin / 8 of XCD c = lin % 8 is (z, piece p) with group (c - p) mod 8 --
    // all task slots of a (piece, group) on ONE XCD, whose L2 then holds that group's source planes once (slots spread over the XCDs:
    // C2-sized smooth case 139 against 91 us).  Pieces without slabs of their own (the pool ran out: one workgroup per channel group
    // out of an emergency slab) keep the static assignment: piece slot lin % sink_x, group lin / sink_x, pieces q = slot, slot + sink_x, ...
    // The grid is as large as the chip holds workgroups of this kernel (launch_scan) and the workgroups DRAW their entries: one counter
    // per list (totals[16 + XCD]; zeroed by scan_box_kernel), so a launch with little work is a few hundred short-lived workgroups, not
    // thousands queueing for the slots behind the tasks, a second task starts on its workgroup's slot without a hand-over, and the
    // workgroups whose task was short (a piece's last one) take what is left.  (Static shares lin, lin + grid, ...: C2-sized smooth case
    // 84 against 79 us before -- 40 workgroups ran two full tasks one after the other.)
    // (State kept across a task: the round, the entries of the rounds before it, my entry index -- the kernel sits at its register
    //  limit: the pieces' task slots are read again for every entry (kept in a register across the task, or in LDS behind a round
    //  marker, the kernel spilled to scratch), R, the list shares etc. are recomputed where they are used.)
    constexpr uint32_t PHASE_B = 0xffffffffu, DRAW = 0xffffffffu;      // step == PHASE_B: the static part; jd == DRAW: no entry index in hand
    uint32_t step = 0, jbase = 0, jd = (a.sink_groups == 8u && (gridDim.x & 7u) == 0u) ? lin >> 3 : lin;
    for (;;) {
        uint32_t q, bslot, g, eslot = 0;
        const uint32_t R = min(32u, a.sink_x);
        if (step != PHASE_B) {
            const bool xcd_lists = a.sink_groups == 8u && (gridDim.x & 7u) == 0u;
            if (step >= (ndef + R - 1u) / R) { step = PHASE_B; jd = lin; jbase = 0; continue; }     // (below: jd = the static index, jbase = its pieces done)
            if (jd == DRAW) {
                // entry lin (>> 3) was mine without a draw; the draws hand out entries share, share + 1, ... of my list
                __syncthreads();
                if (tid == 0) next_j = (xcd_lists ? gridDim.x >> 3 : gridDim.x) + atomicAdd(f.totals + 16 + (xcd_lists ? lin & 7u : 0u), 1u);
                __syncthreads();
                jd = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_j);
            }
            const uint32_t lane = (uint32_t)tid & 63u, qb = step * R, ql = qb + lane;
            uint32_t ns = 0;                               // task slots of the round's pieces, one lane each (0: no slabs of its own)
            if (lane < R && ql < ndef) {
                const uint32_t *h = a.sink_cnt + (size_t)ql * 16u;
                if (h[11] != 0xffffffffu) {
                    const uint32_t nt_ = h[10] != 0xffffffffu ? (h[9] + (uint32_t)Cfg::SEG - 1u) / (uint32_t)Cfg::SEG : h[8];
                    ns = min(a.sink_t, max(nt_, 1u));
                }
            }
            uint32_t j = jd - jbase;                       // (draws only grow: the rounds before this one stay behind)
            uint32_t found = 0xffffffffu, fz = 0, fg = 0, len = 0;
            for (uint32_t z = 0; z < SINK_T; ++z) {
                const uint32_t m = (uint32_t)__ballot(ns > z);
                const uint32_t n = (uint32_t)__popc(m);
                if (n == 0u) break;                        // (the masks are nested: nothing above either)
                const uint32_t per = xcd_lists ? n : n * a.sink_groups;
                len += per;
                if (found == 0xffffffffu && j < per) {
                    const uint32_t r = xcd_lists ? j : j % n;
                    const uint32_t rank = (uint32_t)__popc(m & ((1u << (lane & 31u)) - 1u));
                    const unsigned long long hit = __ballot(lane < 32u && ((m >> (lane & 31u)) & 1u) && rank == r);
                    found = (uint32_t)__ffsll((long long)hit) - 1u;
                    fz = z; fg = xcd_lists ? ((lin & 7u) - found) & 7u : j / n;
                }
                j -= min(j, per);
            }
            if (found == 0xffffffffu) { ++step; jbase += len; continue; }    // (past this round's list: on to the next round with the same entry index)
            q = qb + found; bslot = fz; g = fg;
            jd = DRAW;
            // the usual launch -- one round of pieces, fewer entries than workgroups: nothing to draw after this one
            if (step + 1u == (ndef + R - 1u) / R && jbase + len <= (xcd_lists ? gridDim.x >> 3 : gridDim.x)) step = 0xfffffff0u;
        } else {
            if (jd >= a.sink_x * a.sink_groups) break;
            eslot = jd % a.sink_x; g = jd / a.sink_x; bslot = 0;
            q = eslot + jbase * a.sink_x;
            if (q >= ndef) { jd += gridDim.x; jbase = 0; continue; }
            ++jbase;
            if (a.sink_cnt[(size_t)q * 16u + 11u] != 0xffffffffu) continue;     // (has slabs of its own: rendered above)
        }
        q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q); bslot = (uint32_t)__builtin_amdgcn_readfirstlane((int)bslot); g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
        int cb, ce;
        if (!channel_range((int)s.C, (int)a.sink_groups, (int)g, cb, ce)) continue;    // (arrivals are counted per channel group)
        const size_t slot_floats = (size_t)((uint32_t)s.C + a.sink_groups) * TILE_PIX, my_rows = (size_t)(cb + (int)g) * TILE_PIX;   // slab rows of one task slot over all channel groups; a group's start at cb + its index
        const uint32_t *hd = a.sink_cnt + (size_t)q * 16u;
        const uint32_t sbase = hd[11];
        const bool pooled = sbase != 0xffffffffu;
        const uint32_t all = hd[9], eoff = hd[10];
        const bool listed = eoff != 0xffffffffu;           // the piece's entries were written out: tasks of exactly SEG entries, nothing to walk
        const uint32_t ntask_q = listed ? (all + (uint32_t)Cfg::SEG - 1u) / (uint32_t)Cfg::SEG : hd[8];
        const uint32_t nslot = pooled ? min(a.sink_t, max(ntask_q, 1u)) : 1u;       // (no more slots than tasks)
        float *slab0 = a.sink_pool + (size_t)(pooled ? sbase : a.sink_cap + eslot) * slot_floats + my_rows;
        float *slab = slab0 + (size_t)bslot * slot_floats;
        const uint32_t w = f.defer[q];
        ItemDesc it = {};
        it.tile = w & 0xfffffu; it.nseg = 1u << ((w >> 23) & 3u); it.seg = (w >> 20) & 7u;
        Piece p = make_piece<Cfg>(s, it);
        const rsrc_t rin = sample_planes(s, p, k.hw4);
        PixelSums sums = {0.0f, 0.0f, 0.0f};
        bool wrote = false;
        // One task loop for both kinds of piece (one copy of the record / gather code): listed -- task k_ = entries [k_ * SEG, ...) of the
        // piece's entry array, loaded as they are; not listed -- task = the next pair of candidate source tiles dealt to this slot, walked.
        if (listed) { SINK_STAMP(1, wall_clock64()); SINK_STAMP(8, all); SINK_STAMP(9, (ntask_q - bslot + nslot - 1u) / nslot); }
        uint32_t k_ = bslot;
        int base = 0;
        uint32_t j = 0, n_mine = 0, tk0 = 0;              // pairs: my tasks of the current candidate block; tasks of the blocks before it
        for (;;) {
            uint32_t total;
            if (listed) {
                if (k_ >= ntask_q) break;
                const uint32_t lo = k_ * (uint32_t)Cfg::SEG;
                total = min((uint32_t)Cfg::SEG, all - lo);
                const float4 *src = a.sink_ent + eoff + lo;
                L.cnt[tid] = 0;
#pragma unroll
                for (int i = 0; i < Cfg::EPT; ++i)
                    if ((uint32_t)tid + (uint32_t)i * TT < total) L.ent4[tid + i * TT] = src[tid + i * TT];
                k_ += nslot;
                __syncthreads();
            } else {
                while (j >= n_mine && base < s.tiles) {    // the next block of 2048 source tiles: its candidates, my pairs of them
                    __syncthreads();
                    const uint32_t nc = scan_candidates<Cfg>(s, f, L, p, tid, base);
                    const uint32_t ntask = (nc + 1u) / 2u;
                    const uint32_t t0 = (bslot + nslot - tk0 % nslot) % nslot;
                    n_mine = t0 < ntask ? (ntask - t0 + nslot - 1u) / nslot : 0u;
                    for (uint32_t i = (uint32_t)tid; i < 2u * n_mine; i += TT) {
                        const uint32_t c = 2u * (t0 + (i >> 1) * nslot) + (i & 1u);
                        mine[i] = c < nc ? clist[c] : 0xffffffffu;
                    }
                    tk0 += ntask; j = 0; base += 2048;
                    __syncthreads();
                }
                if (j >= n_mine) break;
                L.cnt[tid] = 0;
                if (tid == 0) L.misc[0] = 0;
                if (tid < 2 * TILE_H) {
                    const uint32_t st = mine[2u * j + (uint32_t)tid / TILE_H];
                    uint32_t w0 = 0u, w1 = 0u;
                    if (st != 0xffffffffu) scan_row_words(s, st, tid % TILE_H, w0, w1);
                    L.rl[tid] = w0; L.rl[ROW_CAP + tid] = w1;
                }
                ++j;
                __syncthreads();
                p.len0 = p.n0 = 2 * TILE_H;
                rows_walk<Cfg, 1, true, false>(s, f, L, p, tid, 0u, 0u, (uint32_t)Cfg::SEG);
                __syncthreads();
                total = L.misc[0];                         // <= 16 row segments x 64 pixels = SEG
            }
            if (total != 0u) {
                EntryRegs<Cfg> e;
                float preA[Cfg::EPT][4], preB[Cfg::EPT][4];
                SINK_STAMP(5, wall_clock64());
                build_records<Cfg, NORM, false>(s, L, p, tid, total, rin, k.hw4, cb, ce - 1, k.shift, k.sc0, k.sc1, e, preA, preB);
                SINK_STAMP(6, wall_clock64());
                stream_planes<Cfg, NORM, MAXOP, false, true, true>(s, f, L, p, tid, rin, k.hw4, cb, ce, e, preA, preB, sums, !wrote, false, slab);
                SINK_STAMP(7, wall_clock64());
                wrote = true;
            }
            __syncthreads();
        }
        // arrive: this workgroup's slab stores (sc1: written through) have been performed; the last workgroup of the (piece, channel
        // group) adds the slabs up with agent-scope loads.  (No fences: an agent-scope fence writes back and invalidates the whole L2 of
        // the XCD -- with 2048 workgroups doing that the sink launch took 440 us.)
        SINK_STAMP(2, wall_clock64());
        __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0)
        __syncthreads();
        if (tid == 0) {
            const uint32_t mineb = wrote ? 1u << (8u + bslot) : 0u;
            arrived = atomicAdd(a.sink_cnt + (size_t)q * 16u + g, 1u | mineb) | mineb;
        }
        __syncthreads();
        const uint32_t aw = arrived;
        SINK_STAMP(3, wall_clock64()); SINK_STAMP(10, (aw & 0xffu) == nslot - 1u); SINK_STAMP(11, q);
        if ((aw & 0xffu) == nslot - 1u) {
            const uint32_t have = aw >> 8;
            const int ly = tid / TILE_W, lx = p.pca + (tid - ly * TILE_W);
            const int oy = p.ty0 + ly, ox = p.tx0 + lx;
            const bool inside = (oy < s.H) & (ox < s.W) & (lx < p.pcb);
            const uint32_t nrow = (uint32_t)(ce - cb), voff = inside ? (uint32_t)(ly * TILE_W + lx) * 4u : BUF_OOB;
            const rsrc_t rs = make_rsrc(slab0, (uint32_t)((nslot - 1u) * slot_floats + (size_t)(nrow + 1u) * TILE_PIX) * 4u);
            float inv = 1.0f;
            // the slabs' values of RC rows in flight together, NS slabs at a time (a load chain per slab and row group is a memory round trip
            // each: 14 slabs x 3 row groups took 13-23 us; sc1 loads come from memory: ~2.5 us a trip)
            auto soff_of = [&](uint32_t z, uint32_t row) { return (uint32_t)(z * slot_floats + (size_t)row * TILE_PIX) * 4u; };
            if (NORM) {
                float tn[SINK_T];
#pragma unroll
                for (uint32_t z = 0; z < SINK_T; ++z) tn[z] = buf_ld_sc1(rs, ((have >> z) & 1u) ? voff : BUF_OOB, soff_of(z, nrow));
                float nrm = 0.0f;
#pragma unroll
                for (uint32_t z = 0; z < SINK_T; ++z) nrm += ((have >> z) & 1u) ? tn[z] : 0.0f;
                inv = 1.0f / norm_divisor(nrm, s.norm_mode, s.eps);
            }
            float *o = f.out + ((size_t)p.n * s.Cs + cb) * hw + (size_t)oy * s.W + ox;
            auto add_up = [&](auto ns_tag, auto rc_tag) {
                constexpr uint32_t NS = decltype(ns_tag)::value, RC = decltype(rc_tag)::value;
                for (uint32_t c = 0; c < nrow; c += RC) {
                    float v[RC];
#pragma unroll
                    for (uint32_t u = 0; u < RC; ++u) v[u] = MAXOP ? s.init : 0.0f;
                    for (uint32_t hv = have; hv;) {                       // NS slabs (in slot order) per trip
                        float t[NS][RC];
                        uint32_t hq = hv;
#pragma unroll
                        for (uint32_t k = 0; k < NS; ++k) {
                            const bool on = hq != 0u;
                            const uint32_t z = on ? (uint32_t)__ffs((int)hq) - 1u : 0u;
                            hq &= hq - 1u;
#pragma unroll
                            for (uint32_t u = 0; u < RC; ++u) t[k][u] = buf_ld_sc1(rs, on ? voff : BUF_OOB, soff_of(z, min(c + u, nrow - 1u)));
                        }
#pragma unroll
                        for (uint32_t k = 0; k < NS; ++k) {
                            const bool on = hv != 0u;
                            hv &= hv - 1u;
#pragma unroll
                            for (uint32_t u = 0; u < RC; ++u)
                                if (on) v[u] = MAXOP ? fmaxf(v[u], t[k][u]) : v[u] + t[k][u];
                        }
                    }
#pragma unroll
                    for (uint32_t u = 0; u < RC; ++u)
                        if (inside && c + u < nrow) __builtin_nontemporal_store(v[u] * inv, o + (size_t)(c + u) * hw);
                }
            };
            if (__popc(have) <= 4) add_up(std::integral_constant<uint32_t, 4>{}, std::integral_constant<uint32_t, 12>{});
            else add_up(std::integral_constant<uint32_t, 8>{}, std::integral_constant<uint32_t, 6>{});
        }
        __syncthreads();
        SINK_STAMP(4, wall_clock64());
    }
    // (the deferred list is emptied by the next call's scan_box_kernel: the scan front end has no prebinned form)
}

// =========================================================================== small kernels

__device__ __forceinline__ float finish(float s, float nrm, int norm_mode, float eps) {
    if (norm_mode == SLR_NORM_ZERO_TO_ONE) return s / (nrm == 0.0f ? 1.0f : nrm);   // softsplat.py:684-686
    return s / fmaxf(nrm, eps);                                                      // ...splating.py:923-924
}


// accum [N,C+1,H,W] (last channel = normaliser) -> out [N,C,H,W]
__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ accum, float *__restrict__ out,
                                                        int C, int HW, int norm_mode, float eps) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *ap = accum + (size_t)n * (C + 1) * HW;
    float *op = out + (size_t)n * C * HW;
    const float nrm = ap[(size_t)C * HW + i];
    for (int c = 0; c < C; ++c) op[(size_t)c * HW + i] = finish(ap[(size_t)c * HW + i], nrm, norm_mode, eps);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}

// two-stage max: stage 1 grid-stride -> partial[blocks]; stage 2 single block -> result[0]
__global__ __launch_bounds__(256) void max_stage_kernel(const float *__restrict__ x, size_t n,
                                                        float *__restrict__ dst) {
    __shared__ float wm[4];
    float m = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) dst[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}


// =========================================================================== host side

int op_ws_open(OpWs &w, int N, int H, int W, void *ws, size_t bytes, const char *who) {
    w.L = op_layout(N, H, W);
    if (!ws || ((uintptr_t)ws & 15) || bytes < w.L.total) {
        set_error("%s: workspace needs %zu bytes (16-byte aligned), got %zu", who, w.L.total, bytes);
        return SLR_E_WORKSPACE;
    }
    char *b = w.base = (char *)ws;
    w.rowcnt = (unsigned long long *)(b + w.L.off_rowcnt);
    w.rowinfo = (unsigned long long *)(b + w.L.off_rowinfo);
    w.rowlist = (RowRec *)(b + w.L.off_rowlist);
    w.rowlist2 = (RowRec *)(b + w.L.off_rowlist2);
    w.items = (ItemDesc *)(b + w.L.off_items);
    w.items2 = (ItemDesc *)(b + w.L.off_items2);
    w.totals = (uint32_t *)(b + w.L.off_totals);
    w.defer = (uint32_t *)(b + w.L.off_defer);
    w.defer2 = (uint32_t *)(b + w.L.off_defer2);
    w.ctl = (uint32_t *)(b + w.L.off_ctl);
    w.arrive = (uint32_t *)(b + w.L.off_arrive);
    w.sink_cnt = (uint32_t *)(b + w.L.off_sink_cnt);
    w.sink_pool = (float *)(b + w.L.off_sink_pool);
    w.sink_ent = (float4 *)(b + w.L.off_sink_ent);
    w.box = b + w.L.off_box;
    return 0;
}

int op_check_dims(int N, int C, int H, int W, const char *who) {
    // image rows travel in 24 bits of a row-list entry, pixel indices in 29 bits; 8 planes of a sample fit one buffer descriptor (a
    // larger plane stack is rendered plane group by plane group: plane_group)
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || H >= (1 << 24) || (long long)N * H * W >= (1LL << 29) || (long long)H * W >= (1LL << 26)) {
        set_error("%s: bad sizes N=%d C=%d H=%d W=%d (N*H*W < 2^29, H*W < 2^26)", who, N, C, H, W);
        return SLR_E_BADARG;
    }
    return 0;
}

// Planes of a sample one launch covers.  A launch addresses the planes of a sample through ONE buffer descriptor (32-bit byte offsets; the
// offset 2^31 is the one no descriptor covers: dropped stores).  The reference indexes up to 2^31 ELEMENTS per tensor (softsplat.py:163,
// 408-416: `int` indices over output.nelement()); a stack of C * H * W * 4 >= 2^31 bytes -- 65 planes of a 4K frame -- is rendered by
// several launches of the tile kernel, each over a group of planes (a multiple of 8, the channel groups' unit) of at most 2^31 - 1
// bytes: the binning / plan is made once, the records of a tile are rebuilt per group.
int plane_group(int C, int H, int W) {
    const long long plane = (long long)H * W * 4, fit = ((1LL << 31) - 1) / plane;
    if (fit >= C) return C;
    return (int)(fit / 8 * 8);                          // (>= 8: H * W < 2^26, op_check_dims)
}

static std::atomic<int> g_scan_max_tiles{SLR_SCAN_MAX_TILES};          // slr_splat_set_scan_max_tiles
static std::atomic<int> g_front_end{SLR_FRONT_END};                    // slr_splat_set_front_end

// Front end of a one-flow call: 1 scan (boxes), 2 rows.  By default the grid decides: scan up to slr_splat_set_scan_max_tiles tiles
// (single-round grids: no plan to wait for), rows above.  A prebinned workspace holds row lists: rows.
static int front_end(int ws_flags, uint32_t nt) {
    if (ws_flags & SLR_WS_PREBINNED) return 2;
    int fe = g_front_end.load(std::memory_order_relaxed);
    if (fe != 1 && fe != 2) fe = nt <= (uint32_t)g_scan_max_tiles.load(std::memory_order_relaxed) ? 1 : 2;
    if (nt >= (1u << 20)) fe = 2;                        // (the scan front end's deferred list carries tile indices in 20 bits)
    return fe;
}

static uint32_t channel_groups(uint32_t nt, int C) {
    if (SLR_CSPLIT_MAX <= 1) return 1u;
    const uint32_t fit = SLR_CSPLIT_SLOTS / (nt ? nt : 1u), byc = (uint32_t)C / 8u;
    uint32_t groups = fit < (uint32_t)SLR_CSPLIT_MAX ? fit : (uint32_t)SLR_CSPLIT_MAX;
    groups = groups < byc ? groups : byc;
    return groups < 1u ? 1u : groups;
}

// scan front end: column pieces x channel groups per tile (slr_splat_set_scan_shape; 0 = by grid size)
static std::atomic<int> g_scan_pieces{0}, g_scan_groups{0}, g_scan_defer_wg{0}, g_scan_defer_groups{0};
static void scan_shape(uint32_t nt, int C, uint32_t &pieces, uint32_t &groups) {
    const int sp = g_scan_pieces.load(), sg = g_scan_groups.load();
    pieces = sp == 1 || sp == 2 || sp == 4 || sp == 8 ? (uint32_t)sp : 1u;
    groups = sg > 0 ? (uint32_t)sg : channel_groups(nt * pieces, C);
    const uint32_t byc = (uint32_t)C / 8u < 1u ? 1u : (uint32_t)C / 8u;
    groups = groups < byc ? groups : byc;
}

// rowbin + plan of one flow into its workspace (the binning of slr_splat_bin and of every self-contained rows call).
static int do_rowbin(const float *flow, OpWs &w, int N, int H, int W, bool clean, hipStream_t st) {
    const uint32_t nt = w.L.nt;
    if (!clean) hipLaunchKernelGGL(rows_zero_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, w.rowcnt, nt, w.ctl, w.arrive);
    const uint32_t grid = (uint32_t)N * (uint32_t)w.L.tiles_x * (uint32_t)((w.L.tiles_y + ROWBIN_R - 1) / ROWBIN_R);
    hipLaunchKernelGGL(rowbin_kernel, dim3(grid), dim3(TILE_PIX), 0, st, flow, w.rowcnt, w.rowinfo, w.rowlist, H, W, w.L.tiles_x, w.L.tiles_y, nt,
                       w.ctl, w.arrive, (uint32_t)OP_SEG, nt > 512u ? (uint32_t)SLR_PLAN_HEAVY : 0u, w.L.items_cap, w.items, w.totals);
    SLR_CHECK_LAUNCH();
    return 0;
}

template <bool NORM, bool MAXOP>
static int launch_rows(OpArgs &a, OpWs &w, hipStream_t st) {
    static LdsOptIn attr_main, attr_pass;
    if (int e = lds_opt_in((const void *)op_rows_kernel<NORM, MAXOP, false>, 159 * 1024, attr_main)) return e;
    if (int e = lds_opt_in((const void *)op_rows_kernel<NORM, MAXOP, true>, 159 * 1024, attr_pass)) return e;
    a.f.rowlist[0] = w.rowlist; a.f.items = w.items; a.f.totals = w.totals; a.f.defer = w.defer; a.f.items_cap = w.L.items_cap;
    // (the grid covers the bound on the plan's items; workgroups past totals[0] exit at once: measured free)
    const uint32_t grid = ((w.L.items_cap + 8 * SLR_XCD_GROUP - 1) / (8 * SLR_XCD_GROUP)) * 8 * SLR_XCD_GROUP;
    const uint32_t groups = channel_groups(w.L.nt, a.s.C);
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));      // slr_splat_time_next: the dominant kernel only
    hipLaunchKernelGGL((op_rows_kernel<NORM, MAXOP, false>), dim3(grid, groups), dim3(TT), OpCfg::LDS_BYTES, st, a);
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
    g_ev_start = g_ev_stop = nullptr;
    // pieces that hold more than SEG entries (none for ordinary flows; appended by their workgroups above): pass by pass, their planes
    // dealt to up to 8 workgroups each (these run after everybody else, on an empty chip)
    const uint32_t wgroups = (uint32_t)a.s.C / 8u < 1u ? 1u : (uint32_t)a.s.C / 8u > 8u ? 8u : (uint32_t)a.s.C / 8u;
    hipLaunchKernelGGL((op_rows_kernel<NORM, MAXOP, true>), dim3(w.L.nt < OP_DEFER_WG ? w.L.nt : OP_DEFER_WG, wgroups), dim3(TT), OpPassCfg::LDS_BYTES, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}

template <bool NORM, bool MAXOP>
static int launch_scan(OpArgs &a, OpWs &w, hipStream_t st) {
    static LdsOptIn attr, attr_s;
    if (int e = lds_opt_in((const void *)op_scan_kernel<NORM, MAXOP>, 159 * 1024, attr)) return e;
    if (int e = lds_opt_in((const void *)op_sink_kernel<NORM, MAXOP>, (int)ScanCfg::LDS_BYTES, attr_s)) return e;
    a.f.box = (const SrcBox *)w.box; a.f.totals = w.totals; a.f.defer = w.defer;
    a.sink_cnt = w.sink_cnt; a.sink_pool = w.sink_pool; a.sink_ent = w.sink_ent; a.sink_ent_cap = w.L.sink_ent_cap;
    // the sink launch's shape: pieces x channel groups x task slots; slabs: rows x 2 KiB per task slot, the emergency slabs of the sink_x piece
    // slots first out of the pool's budget, the rest handed out to the deferred pieces by their own workgroups (the tile kernel below)
    uint32_t pieces, groups;
    scan_shape(w.L.nt, a.s.C, pieces, groups);
    const int dwg = g_scan_defer_wg.load(), dgr = g_scan_defer_groups.load();
    const uint32_t gmax = dgr > 0 && dgr <= 8 ? (uint32_t)dgr : (uint32_t)SLR_SINK_GROUPS;      // (a deferred piece's header counts the arrivals of at most 8 channel groups)
    const uint32_t wgroups = (uint32_t)a.s.C / 8u < 1u ? 1u : (uint32_t)a.s.C / 8u > gmax ? gmax : (uint32_t)a.s.C / 8u;
    const uint32_t dw = dwg > 0 ? (uint32_t)dwg : SINK_P;
    uint32_t sink_x = w.L.nt * pieces < dw ? w.L.nt * pieces : dw;
    const size_t slot_bytes = (size_t)((uint32_t)a.s.C + wgroups) * TILE_PIX * 4, avail = w.L.sink_pool_bytes / slot_bytes;
    SLR_CHECK_ARG(avail >= 1, "plane count too large for the scan front end (use the rows front end: slr_splat_set_front_end(2))");
    if (avail < sink_x) sink_x = (uint32_t)avail;
    a.sink_cap = (uint32_t)(avail - sink_x);
    a.sink_t = a.sink_cap < SINK_T ? a.sink_cap : SINK_T;
    hipLaunchKernelGGL(scan_box_kernel, dim3(w.L.nt), dim3(TILE_PIX), 0, st, a.f.flow[0], (SrcBox *)w.box, a.s.H, a.s.W, w.L.tiles_x, w.L.tiles, w.totals);
    const uint32_t grid = ((w.L.nt + 8 * SLR_XCD_GROUP - 1) / (8 * SLR_XCD_GROUP)) * 8 * SLR_XCD_GROUP;
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));
    hipLaunchKernelGGL((op_scan_kernel<NORM, MAXOP>), dim3(grid, groups, pieces), dim3(TT), ScanCfg::LDS_BYTES, st, a);
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
    g_ev_start = g_ev_stop = nullptr;
    // pieces of more than SEG entries (appended by their workgroups): the sink launch -- every piece by up to SINK_T x channel groups
    // workgroups at once (an empty launch: every workgroup does one scalar load and ends -- 256 or 4096 of them cost the same 2 - 3 us)
    a.sink_x = sink_x; a.sink_groups = wgroups;
    const uint32_t sink_all = sink_x * wgroups * SINK_T;
    hipLaunchKernelGGL((op_sink_kernel<NORM, MAXOP>), dim3(sink_all < (uint32_t)SLR_SINK_GRID ? sink_all : (uint32_t)SLR_SINK_GRID), dim3(TT), ScanCfg::LDS_BYTES, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}

// One call of the operator: front end by grid size / flags, then the tile kernel(s) -- once per plane group (plane_group: one group unless
// the sample's plane stack reaches 2 GiB).
template <bool NORM, bool MAXOP>
static int do_op(OpArgs &a, int ws_flags, void *ws, size_t ws_bytes, hipStream_t st, const char *who) {
    OpWs w;
    if (int e = op_ws_open(w, a.s.N, a.s.H, a.s.W, ws, ws_bytes, who)) return e;
    a.s.tiles_x = w.L.tiles_x; a.s.tiles = w.L.tiles;
    a.f.scale[0] = 1.0f; a.f.scale[1] = 0.0f;
#ifdef SLR_TRACE
    a.s.trace = g_trace;
#endif
    const int fe = front_end(ws_flags, w.L.nt);
    if (fe != 1 && !(ws_flags & SLR_WS_PREBINNED))
        if (int e = do_rowbin(a.f.flow[0], w, a.s.N, a.s.H, a.s.W, (ws_flags & SLR_WS_CLEAN) != 0, st)) return e;
    const int C = a.s.C, gp = plane_group(C, a.s.H, a.s.W);
    const size_t hw = (size_t)a.s.H * a.s.W;
    const float *in = a.s.in;
    float *out = a.f.out;
    a.s.Cs = C;
    for (int pb = 0; pb < C; pb += gp) {
        a.s.in = in + (size_t)pb * hw; a.f.out = out + (size_t)pb * hw;
        a.s.C = C - pb < gp ? C - pb : gp;
        if (int e = fe == 1 ? launch_scan<NORM, MAXOP>(a, w, st) : launch_rows<NORM, MAXOP>(a, w, st)) return e;
    }
    return 0;
}

}  // namespace slr

using namespace slr;

SLR_EXPORT int slr_abi_version(void) { return SLR_ABI_VERSION; }
SLR_EXPORT const char *slr_last_error(void) { return slr::g_err; }

#ifdef SLR_TRACE
SLR_EXPORT void slr_debug_trace(long long *buf) { g_trace = buf; }
#endif

SLR_EXPORT void slr_splat_time_next(void *ev_start, void *ev_stop) {
    slr::g_ev_start = ev_start;
    slr::g_ev_stop = ev_stop;
}

SLR_EXPORT int slr_splat_set_scan_max_tiles(int max_tiles) {
    return slr::g_scan_max_tiles.exchange(max_tiles < 0 ? 0 : max_tiles);
}

SLR_EXPORT int slr_splat_set_front_end(int front_end) {
    return slr::g_front_end.exchange(front_end == 1 || front_end == 2 ? front_end : -1);
}

SLR_EXPORT void slr_splat_set_scan_shape(int pieces, int groups, int defer_wg, int defer_groups) {
    slr::g_scan_pieces.store(pieces); slr::g_scan_groups.store(groups); slr::g_scan_defer_wg.store(defer_wg); slr::g_scan_defer_groups.store(defer_groups);
}

SLR_EXPORT size_t slr_splat_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return op_layout(N, H, W).total;
}

SLR_EXPORT int slr_splat_workspace_init(void *ws, size_t ws_bytes, int N, int H, int W, void *stream) {
    if (int e = op_check_dims(N, 1, H, W, __func__)) return e;
    OpWs w;
    if (int e = op_ws_open(w, N, H, W, ws, ws_bytes, __func__)) return e;
    // everything in front of the lists (counters, plans): zero.  A kernel of our own, not hipMemsetAsync: a captured memset node on a
    // workspace from torch's graph-private pool made HIP-graph replays fault (tests/test_gpu_parity.py::test_frame_is_graph_capturable)
    const size_t words = w.L.off_rowlist / 8;
    hipLaunchKernelGGL(ws_zero_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (unsigned long long *)ws, words);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_splat_bin(const float *flow, int N, int H, int W, void *ws, size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(flow, "null flow");
    if (int e = op_check_dims(N, 1, H, W, __func__)) return e;
    OpWs w;
    if (int e = op_ws_open(w, N, H, W, ws, ws_bytes, __func__)) return e;
    return do_rowbin(flow, w, N, H, W, false, (hipStream_t)stream);
}

SLR_EXPORT int slr_splat_bin_pair(const float *flow_a, const float *flow_b, int N, int H, int W, void *ws_a,
                                  void *ws_b, size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(flow_a && flow_b, "null flow");
    SLR_CHECK_ARG(ws_a != ws_b, "the two flows need separate workspaces");
    if (int e = slr_splat_bin(flow_a, N, H, W, ws_a, ws_bytes, stream)) return e;
    return slr_splat_bin(flow_b, N, H, W, ws_b, ws_bytes, stream);
}

SLR_EXPORT int slr_softsplat_forward(const float *in, const float *flow, float *out, int N, int C, int H,
                                     int W, void *ws, size_t ws_bytes, int prebinned, void *stream) {
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    if (int e = op_check_dims(N, C, H, W, __func__)) return e;
    OpArgs a = {};
    a.s.in = in; a.f.flow[0] = flow; a.f.out = out;
    a.s.N = N; a.s.C = C; a.s.H = H; a.s.W = W; a.s.mulmode = MUL_ONE;
    return do_op<false, false>(a, prebinned, ws, ws_bytes, (hipStream_t)stream, __func__);
}

SLR_EXPORT int slr_softsplat_mode_forward(const float *in, const float *metric, const float *flow, float *out,
                                          int N, int C, int H, int W, int mode, void *ws, size_t ws_bytes,
                                          int prebinned, void *stream) {
    SLR_CHECK_ARG(mode >= SLR_MODE_SUMMATION && mode <= SLR_MODE_SOFTMAX, "mode");
    if (mode == SLR_MODE_SUMMATION)
        return slr_softsplat_forward(in, flow, out, N, C, H, W, ws, ws_bytes, prebinned, stream);
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    SLR_CHECK_ARG(mode == SLR_MODE_AVERAGE || metric, "metric required for linear/softmax");
    if (int e = op_check_dims(N, C, H, W, __func__)) return e;
    OpArgs a = {};
    a.s.in = in; a.s.mul = metric; a.f.flow[0] = flow; a.f.out = out;
    a.s.N = N; a.s.C = C; a.s.H = H; a.s.W = W;
    a.s.mulmode = mode == SLR_MODE_AVERAGE ? MUL_ONE : mode == SLR_MODE_LINEAR ? MUL_PLANE : MUL_EXP;
    a.s.norm_mode = SLR_NORM_ZERO_TO_ONE;
    return do_op<true, false>(a, prebinned, ws, ws_bytes, (hipStream_t)stream, __func__);
}

SLR_EXPORT int slr_maxsplat_forward(const float *in, const float *flow, float *out, float init, int N, int C,
                                    int H, int W, void *ws, size_t ws_bytes, int prebinned, void *stream) {
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    if (int e = op_check_dims(N, C, H, W, __func__)) return e;
    OpArgs a = {};
    a.s.in = in; a.f.flow[0] = flow; a.f.out = out; a.s.init = init;
    a.s.N = N; a.s.C = C; a.s.H = H; a.s.W = W; a.s.mulmode = MUL_ONE;
    return do_op<false, true>(a, prebinned, ws, ws_bytes, (hipStream_t)stream, __func__);
}

SLR_EXPORT int slr_splat_normalize(const float *accum, float *out, int N, int C, int H, int W, int norm_mode,
                                   float eps, void *stream) {
    SLR_CHECK_ARG(accum && out, "null pointer");
    SLR_CHECK_ARG(norm_mode == SLR_NORM_ZERO_TO_ONE || norm_mode == SLR_NORM_CLAMP_EPS, "norm_mode");
    if (int e = op_check_dims(N, C, H, W, __func__)) return e;
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(normalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, accum, out, C, H * W, norm_mode, eps);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_global_max(const float *x, size_t n, float *result, float *scratch, void *stream) {
    SLR_CHECK_ARG(x && result && scratch && n > 0, "null pointer / empty");
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(max_stage_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    hipLaunchKernelGGL(max_stage_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)scratch, (size_t)blocks, result);
    SLR_CHECK_LAUNCH();
    return 0;
}
