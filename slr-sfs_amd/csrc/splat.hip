// splat.hip -- forward splatting as an OWNER-COMPUTES gather (gfx950, no MFMA, no global atomics).
//
// Replaces models/softsplat.py:157-202 (+ the host glue :390-424, :665-690 and the model-side
// weighting / two-direction accumulation / normalisation of forward_flow).
//
// The reference scatters: one thread per ELEMENT, 4 global fp32 atomicAdds each into a
// pre-zeroed output (softsplat.py:186-199,404) -- C-fold redundant flow/weight math, a memset
// and an atomic read-modify-write of every output line on top of the algorithmic traffic.
//
// Here the scatter is turned around.  The flow is shared by all C channels, so it is cheap
// to sort it once:
//   1. bin   (count -> scan -> fill): every source pixel is appended to the bin of each
//            8x64 OUTPUT tile its 2x2 bilinear footprint touches (<= 4 tiles);
//   2. plan : bins are cut into segments of <= SEG entries (load balance: Euler-integrated
//            fluid flows pile up to 7x the average into some tiles) -> work items;
//   3. splat: one workgroup = one work item (tile, segment), one work-item per output pixel.
//            Phase 1 inverts the scatter inside the tile: integer LDS atomics (once per bin
//            entry, not per channel) build a per-output-pixel list of (entry, weight) records.
//            Phase 2 stages the segment's source values in LDS chunk by chunk (coalesced plane
//            loads in bin order, prefetched two chunks ahead) and every work-item accumulates its
//            own pixel in registers.  A single-segment tile is normalised in registers and
//            written with coalesced stores: every output byte is written exactly once, never
//            read, never zeroed.  (details at splat_tile_kernel)
//   4. combine: the few multi-segment tiles wrote raw partial tiles instead; they are summed
//            in segment order, normalised and stored.
// HBM traffic ~ C input planes read (x1.14 bin halo) + C output planes written + 4 B per bin entry,
// i.e. about the algorithmic bytes, instead of ~2x that for the atomic formulation.
#include "slr_common.hpp"
#include "splat_types.hpp"

#include <stdarg.h>
#include <atomic>
#include <type_traits>

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

// =========================================================================== binning

// Wave-aggregated append: lanes of the wave that target the same tile take consecutive
// slots in lane order (so a bin stays sorted by source pixel inside every wave's chunk).
// FILL = false: only count.  Must be called by all 64 lanes (tile < 0 = nothing to add).
template <bool FILL>
__device__ __forceinline__ void wave_append(int tile, uint32_t pix, uint32_t *__restrict__ counter,
                                            const uint32_t *__restrict__ listoff,
                                            uint32_t *__restrict__ list) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(tile >= 0);
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int lt = __shfl(tile, leader);
        unsigned long long same = __ballot(tile == lt);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counter[lt], (uint32_t)__popcll(same));
        if (FILL) {
            base = __shfl(base, leader);
            if (tile == lt) {
                uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                list[listoff[lt] + base + rank] = pix;
            }
        }
        todo &= ~same;
    }
}

// grid (ceil(HW / (256*BIN_PPT)), N).  counter = count[] (FILL=false) or cursor[] (FILL=true).
// The kernel is latency-bound (flow load -> footprint -> reservation atomic -> list store), so
// every work-item handles BIN_PPT pixels (256 apart: each wave still sees 64 consecutive
// pixels of a row per step) with all flow loads, then all fast-path reservations, in flight at once.
constexpr int BIN_PPT = 4;
// Up to two flow fields are binned by the same launches (blockIdx.z picks the flow): the forward
// and the backward displacement map of a frame (the kernels are latency-bound, so two flows cost
// about as much as one).
struct BinSet {
    const float *flow[2];
    uint32_t *count[2], *cursor[2], *listoff[2], *list[2];
};
__global__ __launch_bounds__(256) void zero_counts_kernel(BinSet b, uint32_t nt) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < nt) b.count[blockIdx.y][i] = 0;
}

// Wave-aggregated binning: one global atomic per wave, pixel group and footprint slot.  Used for the COUNT pass of a
// clip (nothing comes back from its atomics, so it never waits: 470 us for a clip's 120 maps against 870 with the LDS
// table below), and for entries that overflow that table.
template <bool FILL>
__device__ __forceinline__ void bin_body_wave(const float *__restrict__ flow, uint32_t *__restrict__ counter,
                                         const uint32_t *__restrict__ listoff, uint32_t *__restrict__ list,
                                         int n, int H, int W, int tiles_x, int tiles) {
    const int HW = H * W;
    const int lane = threadIdx.x & 63;
    const float *f = flow + (size_t)n * 2 * HW;
    const int tb = n * tiles;
    int pix[BIN_PPT];
    float fx[BIN_PPT], fy[BIN_PPT];
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p) {
        pix[p] = (blockIdx.x * BIN_PPT + p) * 256 + threadIdx.x;
        const int q = pix[p] < HW ? pix[p] : 0;
        fx[p] = f[q];
        fy[p] = f[HW + q];
    }
    int tile[BIN_PPT][4], leader[BIN_PPT][4], lt[BIN_PPT][4];
    unsigned long long same[BIN_PPT][4];
    uint32_t base[BIN_PPT][4];
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p) {
        TileSet s = {};
        if (pix[p] < HW) {
            const int y = pix[p] / W, x = pix[p] - y * W;
            s = footprint_tiles(make_corners(fx[p], fy[p], x, y), H, W);
        }
        tile[p][0] = (s.vya & s.vxa) ? tb + s.tya * tiles_x + s.txa : -1;
        tile[p][1] = (s.vya & s.vxb) ? tb + s.tya * tiles_x + s.txb : -1;
        tile[p][2] = (s.vyb & s.vxa) ? tb + s.tyb * tiles_x + s.txa : -1;
        tile[p][3] = (s.vyb & s.vxb) ? tb + s.tyb * tiles_x + s.txb : -1;
        // Fast path: the lanes of a wave (64 consecutive pixels of a row) almost always agree on
        // the tile of a footprint slot; one reservation per slot, issued without waiting.
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long todo = __ballot(tile[p][k] >= 0);
            leader[p][k] = todo ? __ffsll((long long)todo) - 1 : 0;
            lt[p][k] = __shfl(tile[p][k], leader[p][k]);
            same[p][k] = todo ? __ballot(tile[p][k] == lt[p][k]) : 0ull;
            base[p][k] = 0;
            if (same[p][k] && lane == leader[p][k])
                base[p][k] = atomicAdd(&counter[lt[p][k]], (uint32_t)__popcll(same[p][k]));
        }
    }
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (FILL && same[p][k]) {
                const uint32_t b = __shfl(base[p][k], leader[p][k]);
                if (tile[p][k] == lt[p][k] && tile[p][k] >= 0)
                    list[listoff[lt[p][k]] + b + (uint32_t)__popcll(same[p][k] & ((1ull << lane) - 1ull))] = (uint32_t)pix[p];
            }
            if (tile[p][k] == lt[p][k]) tile[p][k] = -1;                       // served
            wave_append<FILL>(tile[p][k], (uint32_t)pix[p], counter, listoff, list);   // lanes that disagree (rare)
        }
}

// Workgroup-level aggregation of the bin reservations, for the FILL pass.  The wave-aggregated version above issues
// one RETURNING global atomic per wave, pixel group and footprint slot and then waits for all of them: 1.26 ms for a
// clip's 120 maps whatever else was changed, 25-45 us for the one map of a stand-alone call.  A workgroup's
// 256 x BIN_PPT consecutive pixels land in a handful of tiles, so they are first counted in a small LDS hash table
// (tile -> entries of this workgroup; integer LDS atomics, fast) and ONE global atomic per distinct tile and
// workgroup reserves the whole chunk.  Inside the chunk an entry's place is the value its LDS atomic returned: the
// lanes of one wave instruction (64 consecutive pixels) keep their order, as before.
constexpr int BIN_HASH = 128;                      // table slots (power of two); > distinct tiles of any sane workgroup

// WAVE_COUNT (compile-time): the COUNT pass in the wave-aggregated form.  Its atomics return nothing, so it never waits,
// and with the 120 maps of a clip in one launch it is the faster one (470 vs 870 us).  The ONE map of a stand-alone call
// is bound by something else: every wave adds to the same few hundred tile counters (~500 atomics per counter at
// 256x480, serialised in L2: 23 us); aggregated per workgroup first, they are 16-64x fewer.
template <bool FILL, bool WAVE_COUNT>
__device__ __forceinline__ void bin_body(const float *__restrict__ flow, uint32_t *__restrict__ counter,
                                         const uint32_t *__restrict__ listoff, uint32_t *__restrict__ list,
                                         int n, int H, int W, int tiles_x, int tiles) {
    if (!FILL && WAVE_COUNT) {
        bin_body_wave<false>(flow, counter, listoff, list, n, H, W, tiles_x, tiles);
        return;
    }
    __shared__ int h_key[BIN_HASH];                // tile id, -1 = free
    __shared__ uint32_t h_cnt[BIN_HASH];           // entries of this workgroup for that tile
    __shared__ uint32_t h_dst[BIN_HASH];           // FILL: where this workgroup's chunk starts in the tile's list
    const int HW = H * W;
    const float *f = flow + (size_t)n * 2 * HW;
    const int tb = n * tiles;
    for (int i = threadIdx.x; i < BIN_HASH; i += 256) { h_key[i] = -1; h_cnt[i] = 0; }
    int pix[BIN_PPT];
    float fx[BIN_PPT], fy[BIN_PPT];
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p) {
        pix[p] = (blockIdx.x * BIN_PPT + p) * 256 + threadIdx.x;
        const int q = pix[p] < HW ? pix[p] : 0;
        fx[p] = f[q];
        fy[p] = f[HW + q];
    }
    __syncthreads();
    // (slot << 16 | rank in the workgroup's chunk) per footprint slot; 0xffffffff: nothing; 0xfffffffe: table full ->
    // that entry takes the wave-aggregated path on global memory below
    uint32_t where[BIN_PPT][4];
    int tile[BIN_PPT][4];
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p) {
        TileSet s = {};
        if (pix[p] < HW) {
            const int y = pix[p] / W, x = pix[p] - y * W;
            s = footprint_tiles(make_corners(fx[p], fy[p], x, y), H, W);
        }
        tile[p][0] = (s.vya & s.vxa) ? tb + s.tya * tiles_x + s.txa : -1;
        tile[p][1] = (s.vya & s.vxb) ? tb + s.tya * tiles_x + s.txb : -1;
        tile[p][2] = (s.vyb & s.vxa) ? tb + s.tyb * tiles_x + s.txa : -1;
        tile[p][3] = (s.vyb & s.vxb) ? tb + s.tyb * tiles_x + s.txb : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            where[p][k] = 0xffffffffu;
            const int t = tile[p][k];
            if (t >= 0) {
                uint32_t slot = ((uint32_t)t * 2654435761u) >> (32 - 7);          // 7 = log2(BIN_HASH)
                where[p][k] = 0xfffffffeu;
                for (int probe = 0; probe < BIN_HASH; ++probe) {
                    const int prev = atomicCAS(&h_key[slot], -1, t);
                    if (prev == -1 || prev == t) {
                        where[p][k] = (slot << 16) | atomicAdd(&h_cnt[slot], 1u);  // < 2^16 entries per workgroup
                        break;
                    }
                    slot = (slot + 1) & (BIN_HASH - 1);
                }
            }
        }
    }
    __syncthreads();
    // one reservation per distinct tile of the workgroup
    for (int i = threadIdx.x; i < BIN_HASH; i += 256) {
        const int t = h_key[i];
        if (t >= 0) {
            const uint32_t b = atomicAdd(&counter[t], h_cnt[i]);
            if (FILL) h_dst[i] = listoff[t] + b;
        }
    }
    if (FILL) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < BIN_PPT; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = where[p][k];
                if (w < 0xfffffffeu) list[h_dst[w >> 16] + (w & 0xffffu)] = (uint32_t)pix[p];
            }
    }
    // overflow of the table (a workgroup whose pixels scatter over more than ~100 tiles: huge incoherent flows)
    bool any_over = false;
#pragma unroll
    for (int p = 0; p < BIN_PPT; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) any_over |= where[p][k] == 0xfffffffeu;
    if (__ballot(any_over)) {                                                  // wave-uniform
#pragma unroll
        for (int p = 0; p < BIN_PPT; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                wave_append<FILL>(where[p][k] == 0xfffffffeu ? tile[p][k] : -1, (uint32_t)pix[p], counter, listoff, list);
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void bin_kernel(BinSet b, int H, int W, int tiles_x, int tiles) {
    bin_body<FILL, false>(b.flow[blockIdx.z], FILL ? b.cursor[blockIdx.z] : b.count[blockIdx.z], b.listoff[blockIdx.z],
                   b.list[blockIdx.z], blockIdx.y, H, W, tiles_x, tiles);
}

// Block-wide exclusive scan helper (1024 threads), returns the block total.
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *excl, uint32_t *wsum /*[16]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
        uint32_t s = wsum[w];
        if (w < wid) woff += s;
        total += s;
    }
    __syncthreads();
    *excl = woff + inc - v;
    return total;
}

// listoff = exclusive prefix sum of count (one workgroup of 1024 threads per flow); zeroes cursor.
__global__ __launch_bounds__(1024) void offsets_kernel(BinSet b, uint32_t nt) {
    const uint32_t *__restrict__ count = b.count[blockIdx.x];
    uint32_t *__restrict__ listoff = b.listoff[blockIdx.x];
    uint32_t *__restrict__ cursor = b.cursor[blockIdx.x];
    __shared__ uint32_t wsum[16];
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nt; b0 += 1024) {
        uint32_t t = b0 + threadIdx.x;
        uint32_t v = t < nt ? count[t] : 0;
        uint32_t ex;
        uint32_t tot = block_exscan(v, &ex, wsum);
        if (t < nt) { listoff[t] = run + ex; cursor[t] = 0; }
        run += tot;
    }
    if (threadIdx.x == 0) listoff[nt] = run;
}

constexpr uint32_t PLAN_SX = 4, PLAN_SY = SLR_PLAN_SY;   // super-tile of the work-item order (plan_kernel).  Measured with
// PLAN_SY = 2 / XCD_GROUP = 8: -7 % HBM fetch (764 -> 708 MB per frame) but no time gain (+1 %): the kernel is not
// traffic-bound at this point, so the simpler row-major order stays the default.

// Work plan for splatting with one (count1 == nullptr) or two flows per tile:
// nseg[t] = segments of the concatenated bin, items[] = (tile, segment) work list, partoff[t] =
// first partial slot of a multi-segment tile.  A tile that does not fit into the partial
// budget is left in one piece (correct, just slower).  Single workgroup of 1024 threads.
// (base0 / base1: where the two bins' lists start in list[0] / list[1], added to ItemDesc.off0 / off1)
__device__ __forceinline__ void plan_body(const uint32_t *__restrict__ count0,
                                          const uint32_t *__restrict__ count1,
                                          const uint32_t *__restrict__ listoff0,
                                          const uint32_t *__restrict__ listoff1, uint32_t base0, uint32_t base1, uint32_t heavy,
                                          uint32_t nt, uint32_t tiles_x, uint32_t tiles_y,
                                          uint32_t seg, uint32_t part_slots,
                                          uint32_t *__restrict__ nseg, uint32_t *__restrict__ partoff,
                                          ItemDesc *__restrict__ items, uint32_t *__restrict__ multi,
                                          uint32_t *__restrict__ whole_items, uint32_t *__restrict__ totals) {
    __shared__ uint32_t wsum[16];
    uint32_t run_items = 0, run_parts = 0, run_multi = 0, run_whole = 0;
    // Work items are emitted in the order of "super-tiles" of PLAN_SY x PLAN_SX tiles (not row-major):
    // the tile kernel places XCD_GROUP consecutive items on one XCD, so with PLAN_SY = 2 a tile's row
    // halo (the source row it shares with the tile below / above) is served by that XCD's L2 as well.
    const uint32_t tiles = tiles_x * tiles_y;
    const uint32_t sup_x = (tiles_x + PLAN_SX - 1) / PLAN_SX, sup_y = (tiles_y + PLAN_SY - 1) / PLAN_SY;
    const uint32_t slots_per_sample = sup_x * sup_y * PLAN_SX * PLAN_SY;
    const uint32_t nslots = (nt / tiles) * slots_per_sample;
    const uint32_t nslots_r = (nslots + 1023u) / 1024u * 1024u;        // passes start on a 1024-slot boundary
    // Heavy tiles first: a launch ends when its last workgroup ends, and the tiles that take 2-3x the median (the
    // ridges of an Euler-integrated field: many entries, long record lists) used to start wherever the spatial order
    // put them -- in the last round as often as in the first.  Two passes over the same spatial order: tiles with
    // more than SLR_PLAN_HEAVY / 4 of the mean entry count, then the rest (neighbours stay neighbours inside each).
    // Measured (one-flow tile kernel, 768x1280, C = 65): Euler t=30 149.6 -> 142.4 us, t=59 177.4 -> 168.0, identity and
    // incoherent flows unchanged; threshold 5/4, 6/4, 8/4 of the mean within 2 %.  `heavy` = 0 (the plans of a clip,
    // whose frames share a launch: the tail of a frame is covered by the next frame's head) keeps the one-pass order.
    const uint32_t all_entries = listoff0[nt] + (listoff1 ? listoff1[nt] : 0u);
    const uint32_t heavy_thr = (uint32_t)(((unsigned long long)all_entries * heavy) / (4ull * (nt ? nt : 1u)));
    for (uint32_t pb = 0; pb < (heavy ? 2u : 1u) * nslots_r; pb += 1024) {
        const int pass = (int)(pb / nslots_r);
        const uint32_t b = pb - (uint32_t)pass * nslots_r;
        const uint32_t slot = b + threadIdx.x;
        uint32_t t = nt;                                 // nt = no tile in this slot (ragged edge of the super-tile grid)
        if (slot < nslots) {
            const uint32_t n = slot / slots_per_sample, r = slot - n * slots_per_sample;
            const uint32_t sup = r / (PLAN_SX * PLAN_SY), k = r - sup * (PLAN_SX * PLAN_SY);
            const uint32_t ty = (sup / sup_x) * PLAN_SY + k / PLAN_SX, tx = (sup % sup_x) * PLAN_SX + k % PLAN_SX;
            if (ty < tiles_y && tx < tiles_x) t = n * tiles + ty * tiles_x + tx;
        }
        uint32_t cnt = 0;
        if (t < nt) cnt = count0[t] + (count1 ? count1[t] : 0u);
        if (heavy && t < nt && (cnt > heavy_thr) != (pass == 0)) t = nt;     // not this pass's tile
        uint32_t ns = t < nt ? (cnt > seg ? (cnt + seg - 1) / seg : 1u) : 0u;
        uint32_t pex;
        uint32_t ptot = block_exscan(ns > 1 ? ns : 0u, &pex, wsum);
        const bool whole = ns > 1 && run_parts + pex + ns > part_slots;   // partial-slot budget exhausted:
        if (whole) ns = 1;               // one workgroup walks ALL segments of the tile itself (ItemDesc.nseg = 0)
        // (a dropped tile leaves a hole in the slot numbering; harmless)
        uint32_t mex;                                                  // compact list of multi-segment tiles
        uint32_t mtot = block_exscan(ns > 1 ? 1u : 0u, &mex, wsum);
        if (ns > 1) multi[run_multi + mex] = t;
        run_multi += mtot;
        uint32_t iex;
        uint32_t itot = block_exscan(t < nt ? ns : 0u, &iex, wsum);
        uint32_t wex;                                                  // compact list of whole-tile items
        uint32_t wtot = block_exscan(whole ? 1u : 0u, &wex, wsum);
        if (whole) whole_items[run_whole + wex] = run_items + iex;
        run_whole += wtot;
        if (t < nt) {
            nseg[t] = ns;
            partoff[t] = run_parts + pex;
            ItemDesc d;
            d.tile = t; d.cnt0 = count0[t]; d.cnt1 = count1 ? count1[t] : 0u;
            d.off0 = base0 + listoff0[t]; d.off1 = listoff1 ? base1 + listoff1[t] : 0u;
            d.nseg = whole ? 0u : ns; d.partoff = run_parts + pex;
            for (uint32_t s = 0; s < ns; ++s) { d.seg = s; items[run_items + iex + s] = d; }
        }
        run_items += itot;
        run_parts += ptot;
    }
    if (threadIdx.x == 0) { totals[0] = run_items; totals[1] = run_parts; totals[3] = run_multi; totals[4] = run_whole; }
}

__global__ __launch_bounds__(1024) void plan_kernel(const uint32_t *__restrict__ count0,
                                                    const uint32_t *__restrict__ count1,
                                                    const uint32_t *__restrict__ listoff0,
                                                    const uint32_t *__restrict__ listoff1, uint32_t nt,
                                                    uint32_t tiles_x, uint32_t tiles_y,
                                                    uint32_t seg, uint32_t part_slots,
                                                    uint32_t *__restrict__ nseg, uint32_t *__restrict__ partoff,
                                                    ItemDesc *__restrict__ items, uint32_t *__restrict__ multi,
                                                    uint32_t *__restrict__ whole_items, uint32_t *__restrict__ totals) {
    // (heavy tiles first only where a launch takes more than one round of workgroups: the second pass costs 3.5 us)
    plan_body(count0, count1, listoff0, listoff1, 0u, 0u, nt > 512u ? (uint32_t)SLR_PLAN_HEAVY : 0u, nt, tiles_x, tiles_y, seg, part_slots,
              nseg, partoff, items, multi, whole_items, totals);
}

// =========================================================================== scan front end (no bins)
// Small grids (config C2 of BASELINE.json: 256x480) spend their time in the latency chains of seven tiny dependent
// launches (zero, count, scan, fill, plan, whole, combine: ~35 us around a 30 us tile kernel, profiles/r2_c2_kernel_stats.txt).
// The scan front end replaces all of them by ONE small kernel: every 8x64 block of SOURCE pixels ("source tile") gets
// the bounding box of the NW corners its pixels splat to.  The tile kernel (SCAN instantiation) then needs no bins:
// an output tile's workgroup tests all boxes (16 bytes each, L2-resident), scans the flow of the few source tiles
// whose box touches the tile (wave w = row w of the source tile: coalesced 256-byte loads) and builds its entry list
// in LDS itself.  Any flow is handled exactly: a box that covers everything just means more candidates to scan.
struct SrcBox { int x0, x1, y0, y1; };      // inclusive range of the footprints' NW corners; x0 > x1: no pixel splats into the image

__global__ __launch_bounds__(TILE_PIX) void scan_box_kernel(const float *__restrict__ flow, SrcBox *__restrict__ box,
                                                            int H, int W, int tiles_x, int tiles, uint32_t *__restrict__ ctl,
                                                            unsigned long long *__restrict__ q_items, uint32_t *__restrict__ arrive,
                                                            uint32_t part_slots) {
    const int t = blockIdx.x, n = t / tiles, tl = t - n * tiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // the segment-sharing state of the tile kernel that follows (a kernel boundary makes the zeros visible to it)
    for (uint32_t i = (uint32_t)t * TILE_PIX + tid; i < part_slots; i += gridDim.x * TILE_PIX) { q_items[i] = 0ull; arrive[i] = 0u; }
    if (t == 0 && tid < 64) ctl[tid] = 0u;      // (ctl[3], tickets on offer, is signed: draws may race ahead of the publisher's add)
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

// =========================================================================== rows front end
// The bins cost a stand-alone call 55-95 us around its tile kernel at 768x1280 (zero, count, offsets, fill, plan, whole, combine:
// seven launches, two passes over the flow with one atomic per pixel footprint, 18-34 us of combine); the scan front end above
// has no plan, so it meets a heavy tile only when its workgroup happens to start.  Third form: bin ROW SEGMENTS instead of pixels.
// A row segment = 64 consecutive source pixels of one image row (one wave's coalesced load).  rowbin_kernel reads the flow once;
// every wave finds the few tiles its segment's footprints touch (ballots, no LDS) and appends (segment, hits) to each of them
// with ONE returning 64-bit atomic -- ~35-100 k atomics at 768x1280 instead of ~4 M -- which also sums the tile's exact entry
// count; a second, non-returning atomic adds to the tile's column-octant histogram.  The workgroup that finishes last turns the
// counts into the work plan: heavy tiles first, a tile of more than SEG entries cut into ranges of its OUTPUT COLUMNS -- a piece
// stages every entry that touches its columns and writes its output pixels itself, so there are no partial tiles and nothing to
// combine.  The tile kernel (FE = 2) loads its tile's list (<= SLR_ROW_CAP segments), scans exactly those rows of the flow and
// places the hits at slots known from the list's counts (whole tiles) or handed out by one LDS atomic per row (pieces); it has
// no pass loop and no work loop: 80 VGPRs, three workgroups per CU.  A piece that still holds more than SEG entries is handed
// to a second, normally empty launch (WHOLE) that walks it pass by pass.  A tile touched by more than SLR_ROW_CAP segments
// (pathological flows) is scanned from ALL rows of its sample.  DESIGN.md 3.2.5b has the measurements behind every choice.

// Work plan from the per-tile (entries, row segments) words; run by ONE workgroup of TILE_PIX work-items (the last one of
// rowbin_kernel) in ONE pass over the tiles.  A round covers 4 * TILE_PIX tiles: every work-item loads the words of 4
// CONSECUTIVE tiles together (one memory round trip per round), sums them locally, and two workgroup scans per round -- partial
// slots, then (heavy | other) items packed in one 64-bit word -- place them (a scan is a chain of cross-lane steps and
// barriers, ~0.5 us: one per tile and quantity made the plan 9.5 us of a 32 us kernel at 1920 tiles).
// Tiles in row-major order; on grids of more than one round of workgroups the heavy tiles (more than SLR_PLAN_HEAVY / 4 of an
// undisturbed tile's ~585 entries) go first: their items fill items[] from the front, everybody else's from the back
// (items[cap - 1 - k]), and the tile kernel reads item i < totals[5] from the front.  Segments of `seg` entries; a tile whose
// segments do not fit the partial-slot budget is left to one workgroup (nseg 0).
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

__device__ __forceinline__ void rows_plan(const unsigned long long *__restrict__ rowcnt, uint32_t nt, uint32_t seg, uint32_t heavy,
                                          uint32_t items_cap, ItemDesc *__restrict__ items, uint32_t *__restrict__ totals) {
    __shared__ unsigned long long wsum[TILE_PIX / 64];
    // The words were written by other workgroups' agent-scope atomics (performed at the memory side, before their arrival
    // atomics); this XCD's L2 may still hold the zeros of rows_zero_kernel.  They are read with agent-scope (sc1) loads, all four
    // of a round in flight together (as __hip_atomic_load the compiler waits after each).
    constexpr uint32_t PER = SLR_ROWS_PLAN_PER;
#ifdef SLR_PLAN_STAMPS
#define PSTAMP(k) do { if (threadIdx.x == 0) ((unsigned long long *)totals)[8 + (k)] = wall_clock64(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
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
        for (uint32_t k = 0; k < PER; ++k) if (t0 + k >= nt) w[k] = 0ull;
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
                pcs[k] = p; ns[k] = np;
#if SLR_ROWS_EVEN_FIRST
                // (where plain halves / quarters already fit, take them: equal widths)
                uint32_t q4[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q)
                    q4[q] = (uint32_t)((float)((uint32_t)(oh[k] >> (16 * q)) & 0xffu) * scale) + (uint32_t)((float)((uint32_t)(oh[k] >> (16 * q + 8)) & 0xffu) * scale);
                if (max(q4[0] + q4[1], q4[2] + q4[3]) <= limit) { pcs[k] = 0x40ull | (0x44ull << 8); ns[k] = 2; }
                else if (max(max(q4[0], q4[1]), max(q4[2], q4[3])) <= limit && np >= 4) { pcs[k] = 0x20ull | (0x22ull << 8) | (0x24ull << 16) | (0x26ull << 24); ns[k] = 4; }
#endif
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
    if (threadIdx.x == 0) { totals[0] = run_heavy + run_light; totals[1] = 0; totals[3] = 0; totals[4] = 0; totals[5] = run_heavy; }
}

// grid: N * tiles_x * ceil(tiles_y / ROWBIN_R) workgroups; a workgroup covers ROWBIN_R vertically adjacent source tiles (wave w: rows
// w, w + 8, ... of the block, all their flow loads in flight together, their appends in ONE atomic instruction): a wave's life is
// one load round trip + one atomic round trip however many rows it carries, and 1920 one-tile workgroups took 2.5 rounds of that.
constexpr int ROWBIN_R = SLR_ROWBIN_R;
__global__ __launch_bounds__(TILE_PIX) void rowbin_kernel(const float *__restrict__ flow, unsigned long long *__restrict__ rowcnt,
                                                          RowRec *__restrict__ rowlist, int H, int W, int tiles_x, int tiles_y,
                                                          uint32_t nt, uint32_t *__restrict__ ctl, uint32_t *__restrict__ arrive1,
                                                          uint32_t seg, uint32_t heavy, uint32_t items_cap,
                                                          ItemDesc *__restrict__ items, uint32_t *__restrict__ totals) {
    const int tiles = tiles_x * tiles_y, by_n = (tiles_y + ROWBIN_R - 1) / ROWBIN_R, per_n = tiles_x * by_n;
    const int b = blockIdx.x, n = b / per_n, bl = b - n * per_n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#ifdef SLR_PLAN_STAMPS
    const unsigned long long k_entry = wall_clock64();
    if (b == 0 && tid == 0) ((unsigned long long *)totals)[15] = k_entry;
#endif
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
            const Corners c = make_corners(fx[r], fy[r], x, y);
            const TileSet q = footprint_tiles(c, H, W);
            if (q.vxa & q.vya) t0 = q.tya * tiles_x + q.txa;
            if (q.vxb & q.vya) t1 = q.tya * tiles_x + q.txb;
            if (q.vxa & q.vyb) t2 = q.tyb * tiles_x + q.txa;
            if (q.vxb & q.vyb) t3 = q.tyb * tiles_x + q.txb;
            // tile column a holds corner column x0 (if in the image) and x0 + 1 when it lies in the same tile column; tile column b
            // (valid only when distinct) holds x0 + 1.  Octant = 8 output columns (the finest piece of a heavy tile).
            const bool x0in = c.ok & (c.x0 >= 0) & (c.x0 < W), x1in = c.ok & (c.x0 + 1 >= 0) & (c.x0 + 1 < W);
            if (q.vxa) cm_a = (x0in ? 1u << ((c.x0 & (TILE_W - 1)) >> 3) : 0u) |
                              ((x1in && (c.x0 + 1) / TILE_W == q.txa) ? 1u << (((c.x0 + 1) & (TILE_W - 1)) >> 3) : 0u);
            if (q.vxb) cm_b = 1u << (((c.x0 + 1) & (TILE_W - 1)) >> 3);
        }
        for (;;) {
            const int cand = t0 >= 0 ? t0 : t1 >= 0 ? t1 : t2 >= 0 ? t2 : t3;
            const unsigned long long pend = __ballot(cand >= 0);
            if (!pend) break;
            const int leader = __ffsll((long long)pend) - 1;
            const int T = __builtin_amdgcn_readlane(cand, leader);
            const bool h = (t0 == T) | (t1 == T) | (t2 == T) | (t3 == T);
            const uint32_t c = (uint32_t)__popcll(__ballot(h));
            const uint32_t lm = (((t0 == T) | (t2 == T)) ? cm_a : 0u) | (((t1 == T) | (t3 == T)) ? cm_b : 0u);   // column octants of T this lane touches
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
                for (unsigned long long m = __ballot(h); m; m &= m - 1ull)
                    rm |= (uint32_t)__builtin_amdgcn_readlane((int)lm, __ffsll((long long)m) - 1);
            }
            if (t0 == T) t0 = -1;
            if (t1 == T) t1 = -1;
            if (t2 == T) t2 = -1;
            if (t3 == T) t3 = -1;
            if (lane == k) { my_tile = T; my_cnt = c; my_y = y | (int)(rm << 24); my_hist = hist; }
            if (++k == 64) flush();
        }
    }
    flush();
    // ---- the last workgroup to get here plans the call (every append above has returned: its value was used)
    // (two levels: thousands of returning atomics on ONE word are served one after the other -- 35 us at 1920 workgroups)
    __shared__ uint32_t last;
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
#ifdef SLR_PLAN_STAMPS
    if (tid == 0) { ((unsigned long long *)totals)[14] = k_entry; ((unsigned long long *)totals)[13] = (unsigned long long)wall_clock64(); }
#endif
    rows_plan(rowcnt, nt, seg, heavy, items_cap, items, totals);
}

// (rowcnt words and the arrival counter of rowbin_kernel, zeroed at the start of every call: the workspace is the caller's memory)
__global__ __launch_bounds__(256) void rows_zero_kernel(unsigned long long *__restrict__ rowcnt, uint32_t nt, uint32_t *__restrict__ ctl,
                                                        uint32_t *__restrict__ arrive1) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nt) { rowcnt[2 * (size_t)i] = 0ull; rowcnt[2 * (size_t)i + 1] = 0ull; }
    if (i < (nt + 63u) / 64u) arrive1[(size_t)i * 32u] = 0u;    // first-level arrival counters of rowbin_kernel: one per 64
                                                                 // workgroups, each on its own 128-byte line (atomics on one line are served
                                                                 // one after the other: 1920 arrivals on one line cost 14 us)
    if (i < 16u) ctl[i] = 0u;
}

// =========================================================================== splat

enum { MUL_ONE = 0, MUL_PLANE = 1, MUL_EXP = 2, MUL_EXP_SHIFT = 3 };

struct SplatArgs {
    const float *in;        // [N,C,H,W] value planes
    const float *mul;       // [N,1,H,W] weight plane (metric / Z), or nullptr
    const float *mulmax;    // device scalar subtracted before exp (MUL_EXP_SHIFT)
    const float *flow[2];   // [N,2,H,W] per direction (flow[1] unused when ndir == 1)
    const uint32_t *count[2], *listoff[2], *list[2];
    float scale[2];         // alpha, 1 - alpha
    const uint32_t *nseg, *partoff, *totals, *multi, *whole_items;
    const ItemDesc *items;
    float *partial, *trash;
    float *out;             // [N,C,H,W]
    float *norm_out;        // [N,1,H,W] or nullptr
    size_t part_stride;
    int N, C, H, W, tiles_x, tiles;
    int ndir, seg;
    int mulmode, norm_mode;
    float eps, init;
    long long *trace;
    const SrcBox *box;      // SCAN front end: destination boxes of the source tiles [N * tiles]
    uint32_t nt, part_slots;  // SCAN front end: N * tiles (there is no plan to read a total from); partial-tile slots
    uint32_t *ctl;          // SCAN front end, segment sharing (see ScanCtl): head / tail / slots used; nullptr = off
    unsigned long long *q_items;   // [part_slots] queue of (tile, segment) work, one 8-byte word each, 0 = not written yet
    uint32_t *arrive;       // [part_slots] arrival counter of a shared tile, indexed by its first partial slot
    uint32_t items_cap, pad3_;   // ROWS front end: size of items[] (rows_plan fills it from both ends)
    const RowRec *rowlist;  // ROWS front end: [N * tiles][ROW_CAP] row segments of every tile (rowbin_kernel)
    // SECOND WEIGHT GROUP of the fused two-flow kernel (the 2-layer model's alpha plane: ONE value plane with its own
    // weight plane, ..._2layers_alpha_seperate.py:963-1045), splatted by the same launch with the same records:
    const float *in2;       // [N,1,H,W] or nullptr
    const float *mul2;      // [N,1,H,W] its weight logits: w2 = exp(mul2) (mulmode2 == MUL_EXP) or mul2 (MUL_PLANE)
    float *out2;            // [N,1,H,W] = sum(w * w2 * in2) / max(sum(w * w2), eps)
    int mulmode2, pad2_;
#ifdef SLR_SCAN_ORDER_HOOK
    const uint32_t *order;
#endif
};

// Several frames of a clip in ONE launch: kernels on a stream run one after the other, so with one launch per frame
// the last round of a frame's work items runs on a half-empty chip (2100 items for 512 workgroup slots: 4.1 rounds)
// before the next frame's first round may start; measured with two streams, letting consecutive frames overlap takes
// the splat of a frame from 243 to 190-200 us.  The arguments of up to MAXB frames travel as kernel arguments; a
// workgroup finds its frame from its block index (ranges end[] for the tile kernel, cend[] for combine), wave-uniform.
constexpr int MAXB = SLR_MAXB;
struct SplatBatch {
    SplatArgs f[MAXB];
    uint32_t end[MAXB];      // tile kernel: blocks [end[i-1], end[i]) belong to frame i (multiples of 8 * XCD_GROUP) ...
    uint32_t cend[MAXB];     // combine kernel: the same for its grid.x
    uint32_t nb;
    uint32_t interleave;     // ... or (tile kernel, nb > 1): groups of 8 * XCD_GROUP blocks round-robin over the frames,
                             // end[i] = frame i's OWN grid size (see launch_batch)
};

static_assert(sizeof(SplatBatch) <= 4096, "kernel arguments are limited to 4 KiB");

__device__ __forceinline__ uint32_t batch_frame(const uint32_t (&end)[MAXB], uint32_t nb, uint32_t bx, uint32_t &start) {
    uint32_t f = 0;
    start = 0;
#pragma unroll
    for (int i = 0; i + 1 < MAXB; ++i)
        if (i + 1 < (int)nb && bx >= end[i]) { f = i + 1; start = end[i]; }
    return f;
}

__device__ __forceinline__ float finish(float s, float nrm, int norm_mode, float eps) {
    if (norm_mode == SLR_NORM_ZERO_TO_ONE) return s / (nrm == 0.0f ? 1.0f : nrm);   // softsplat.py:684-686
    return s / fmaxf(nrm, eps);                                                      // ...splating.py:923-924
}

__device__ __forceinline__ float norm_value(float nrm, int norm_mode, float eps) {
    return norm_mode == SLR_NORM_ZERO_TO_ONE ? (nrm == 0.0f ? 1.0f : nrm) : fmaxf(nrm, eps);
}

#ifdef SLR_TRACE      // development aid: per-workgroup phase timestamps (s_memtime) into a.trace
#define SLR_TRACE_SLOTS 64      // 0-3, 28: phase 1; 4..30: the first 9 chunks (3 stamps each); 32..: specials
#define SLR_STAMP(slot) do { if (a.trace && threadIdx.x == 0 && (slot) < SLR_TRACE_SLOTS) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLR_TRACE_SLOTS + (slot)] = clock64(); } while (0)
// (clock64 counters differ between XCDs: chip-wide timelines use the 100 MHz real-time counter)
#define SLR_STAMP_RT(slot) do { if (a.trace && threadIdx.x == 0) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLR_TRACE_SLOTS + (slot)] = wall_clock64(); } while (0)
#else
#define SLR_STAMP(slot) do { } while (0)
#define SLR_STAMP_RT(slot) do { } while (0)
#endif
#ifdef SLR_CUT                                   // measurement builds: the SCAN kernel ends at phase boundary SLR_CUT (time-to-phase)
#define SLR_CUT_AT(k) do { if (SLR_CUT == (k)) return; } while (0)
#else
#define SLR_CUT_AT(k) do { } while (0)
#endif

constexpr int SPLAT_THREADS = TILE_PIX;        // one work-item per output pixel of the tile
constexpr int XCD_GROUP = SLR_XCD_GROUP;                    // neighbouring tiles kept on one XCD (L2 halo reuse)
constexpr int RB = 4;                          // records per batch in the gather loop
constexpr int LMAX = SLR_LMAX;                       // records a work-item walks alone (longer lists: wave-cooperative)
constexpr int COMBINE_CHUNK = 8;               // planes per combine workgroup
// Per-variant shape: EPT bin entries per work-item (segment = EPT*TILE_PIX entries), CHUNK planes
// staged in LDS / accumulated in registers per pass.  LDS per workgroup (6-byte records):
//   one flow  EPT 2, CHUNK 4 -> 3.1 + 27.0 + 16 = 46 KiB
//   two flows EPT 3, CHUNK 4 -> 3.1 + 39.0 + 24 = 66 KiB  -> two workgroups per CU
// Measured (768x1280, C = 65, tile kernel alone, identity / Euler t=30 / t=59 / incoherent flow, us):
//   8-byte records, CHUNK 8 (round 1)   : 152 / 196 / 244 / 228
//   6-byte records, CHUNK 4             : 144 / 178 / 223 / 208
// (6-byte records alone: no change; CHUNK 4 alone: 148 / 188 / 236 / 218.  The register cap does not matter: 4 / 5 / 6
// waves per SIMD = 2 / 2 / 3 workgroups per CU measure the same within 1 %.)
                               // its lists are longer; measured 201 vs 206 us per frame)
constexpr int EPT_ONE = SEG_ONE / SPLAT_THREADS, CHUNK_ONE = SLR_CHUNK_ONE;     // (CHUNK 4 measures the same: the chunk count /
                                                                    // barrier count is not what bounds the kernel)
constexpr int EPT_TWO = SEG_TWO / SPLAT_THREADS, CHUNK_TWO = 4;
// records per segment: <= 4 per entry, + 1 pad per output pixel (list lengths are made odd so the
// lanes' list walks start on different LDS banks: with the typical 4 records per pixel an
// unpadded CSR puts lanes i and i+8 on the same bank -> 8-way conflict on every record read)
__host__ __device__ constexpr int rec_cap(int ept) { return 4 * ept * SPLAT_THREADS + SPLAT_THREADS; }

// 16-byte LDS slot of planes [4h, 4h+4) of staged entry e.  CHUNK 4: one slot per entry, consecutive
// entries -> consecutive slots (conflict-free ds_read_b128).  CHUNK 8: two slots per entry; the half
// is XOR-ed with bit 3 of the entry so that a 16-lane group reading the same half of 16 consecutive
// entries covers all 16 slots of the 256-byte LDS row instead of hitting 8 of them twice.
template <int CHUNK>
__device__ __forceinline__ uint32_t vslot(uint32_t e, int h) {
    static_assert(CHUNK == 4 || CHUNK == 8, "staging layout");
    return CHUNK == 4 ? e : 2u * e + ((uint32_t)h ^ ((e >> 3) & 1u));
}

// Why no LDS float accumulation: on gfx950 ds_add_f32 retires ~1 lane per 2.6 clocks (170
// CU-cycles per wave instruction) and even ds_add_f64 (7.8) made the 4-atomics-per-element
// formulation LDS-bound (tools/ubench/lds_atomic.hip).  Integer LDS atomics are fast (4.5),
// so they are used ONCE per bin entry to invert the scatter inside the tile:
//
//   phase 1  every bin entry of the segment appends (source pixel, weight) records to the
//            per-OUTPUT-pixel lists of its <= 4 in-tile corners: ds_add_rtn_u32 on a count per
//            output pixel, a workgroup scan, ds_write_b64 of the records (a CSR in LDS);
//   phase 2  per chunk of 8 channels: the segment's source values are STAGED in LDS with
//            coalesced plane loads in bin order (each source value is read from HBM/L2 once per
//            tile it touches, prefetched one chunk ahead in registers); then one work-item per
//            output pixel walks its own record list, reads the staged values (ds_read_b32, no
//            atomics) and accumulates v*w IN REGISTERS, and stores its pixel: every output
//            byte is written exactly once, coalesced, never read, never zeroed.
//
// Workgroup = (tile, segment); TILE_PIX threads; LDS = counts + offsets + 4*seg records + 8*seg values.
// bytes of LDS in front of the staged values: counts, wave sums, offsets, records (16-byte aligned)
constexpr int EPT_SCAN = SLR_EPT_SCAN;
__host__ __device__ constexpr bool rec6(int ept, bool scan = false) { return SLR_REC6 && (scan || ept * SPLAT_THREADS == SEG_ONE); }
__host__ __device__ constexpr size_t lds_head_bytes(int ept, bool scan = false) {
    return ((size_t)(SPLAT_THREADS + 16 + SPLAT_THREADS / 2) * 4 + (size_t)rec_cap(ept) * (rec6(ept, scan) ? 6 : 8) + 15) & ~(size_t)15;
}
// (the rare whole-tile instantiation carries a segment loop and would spill under the 80-register cap)
constexpr int tile_min_waves(int ept, bool whole, int fe) {
    return whole ? 1 : fe == 2 ? SLR_WAVES_ROWS : fe == 1 ? SLR_WAVES_SCAN : (ept == EPT_ONE && SLR_WAVES_ONE > 0) ? SLR_WAVES_ONE : 1;
}

// ---- segment sharing inside the SCAN kernel ------------------------------------------------------------------------
// A tile with more than SEG entries (the ridges of an Euler-integrated field hold up to 7x the mean) used to be walked
// pass by pass by its one workgroup while the rest of the chip went idle (768x1280, t = 59: 318 us against 246 with bins
// and a plan).  Without a count pass nobody knows the heavy tiles in advance, so the split is made when they are found:
// the tile's workgroup reserves ns partial-tile slots, PUSHES segments 1..ns-1 as 8-byte work words onto a queue in the
// workspace and takes segment 0 itself.
// Taking work is guarded by a semaphore (ctl[3] = words pushed and not yet claimed): a claim is one returning atomic
// decrement; only a granted claim draws a queue index (fetch-add on the head), so the head can never overshoot the
// tail and there is no compare-and-swap anywhere.  (Measured before it looked like this: a CAS-popped queue collapsed
// under contention, 40 us per pop at t = 59; a list of open tiles with per-tile tickets that every finishing workgroup
// looked at cost 16 us per workgroup end on average -- every step is a dependent memory-side operation behind the
// streaming traffic, and ~50 workgroups end inside the age of one snapshot.)  Who claims:
//   * a workgroup that has just finished a segment (the publisher after segment 0 included): always, on fresh state -- so
//     every pushed word is taken by somebody even if nobody else turns up;
//   * one workgroup in SLR_SHARE_HELPERS after its own tile, and only if the control words it fetched two chunks before
//     the end of its work (arrived by then: no waiting) show words on offer.  The other workgroups never look.
// Each segment's raw sums go to its partial slot with WRITE-THROUGH stores (sc1: straight to memory, no L2 write-back
// fence -- the XCDs' L2s are not coherent), every storing wave drains vmcnt, then one agent-scope atomic counts the arrival;
// the workgroup that arrives last makes ONE agent-scope acquire, reads all ns slots back, sums them in segment order,
// normalises and stores the tile.  No plan, no combine launch, and the only wait on another workgroup is for a queue word
// whose writer has already reserved it (it writes it a few instructions later).
// Control words (zeroed by scan_box_kernel of the same call): ctl[0] queue head, ctl[1] queue tail, ctl[2] partial slots
// used, ctl[3] semaphore (signed: claims may race ahead of a push and are given back).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;

__device__ __forceinline__ void store_wt16(float *p, f4v v) {          // 16-byte write-through store (untracked by the compiler's
#if SLR_SHARE_STORE == 3
    // two 8-byte agent-scope stores the COMPILER issues (global_store_dwordx2 ... sc1) and therefore counts: an inline-asm store is
    // invisible to its vmcnt bookkeeping, so every wait for a prefetched plane also waited for the younger write-through store
    // to be acknowledged by memory -- one write latency per chunk, 17 per piece
    typedef unsigned long long u64a __attribute__((may_alias));
    const u64a lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
    const u64a hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
    __hip_atomic_store((gu64 *)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gu64 *)p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif SLR_SHARE_STORE == 0
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");   // vmcnt bookkeeping: drained by hand)
#elif SLR_SHARE_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<f4v *>(p) = v;
#endif
}
// agent-scope (sc1) loads of what other workgroups stored write-through: they do not trust this XCD's L2.  Issued in batches and
// waited for by hand (as __hip_atomic_load the compiler waits after every single one; an agent-scope acquire fence + plain loads
// invalidates the XCD's whole L2 under everybody else's feet: +26 us per call at 128 multi-piece tiles).
__device__ __forceinline__ f4v load_sc1_16(const float *p) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float load_sc1_4(const float *p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_wt4(float *p, float v) {
    __hip_atomic_store((gu32 *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// work word: [63:38] tile + 1 (never 0), [37:30] segment, [29:22] segments of the tile, [21:4] first partial slot, [3:0] channel group
__device__ __forceinline__ unsigned long long pack_work(uint32_t tile, uint32_t seg, uint32_t ns, uint32_t po, uint32_t grp) {
    return ((unsigned long long)(tile + 1u) << 38) | ((unsigned long long)seg << 30) | ((unsigned long long)ns << 22) |
           ((unsigned long long)po << 4) | grp;
}
constexpr uint32_t SHARE_MAX_SEG = 255u, SHARE_MAX_SLOT = 1u << 18, SHARE_MAX_TILE = (1u << 26) - 2u;

// FE: front end of the launch -- 0 bins (entry lists + plan), 1 scan (boxes, no plan), 2 rows (row-segment lists + plan)
template <bool NORM, bool MAXOP, int EPT_MAX, int CHUNK, bool WHOLE, int FE = 0>
__global__ __launch_bounds__(SPLAT_THREADS, tile_min_waves(EPT_MAX, WHOLE, FE)) void splat_tile_kernel(SplatBatch batch) {
    constexpr bool SCAN = FE != 0;          // the tile's entries are built in LDS from the flow itself (no bin lists)
    constexpr bool ROWS = FE == 2;          // ... from the tile's row-segment list, work items from a plan
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t bstart, bf, bx;                               // frame of this block, block index inside the frame's own grid
    if (batch.interleave) {
        constexpr uint32_t G = 8 * XCD_GROUP;
        const uint32_t gg = blockIdx.x / G;
        bf = gg % batch.nb;
        bx = (gg / batch.nb) * G + blockIdx.x % G;
        if (bx >= batch.end[bf]) return;                   // this frame has fewer groups than the longest of the batch
    } else {
        bf = batch_frame(batch.end, batch.nb, blockIdx.x, bstart);
        bx = blockIdx.x - bstart;
    }
    const SplatArgs &a = batch.f[bf];                      // scalar loads with a dynamic offset
    constexpr int T = SPLAT_THREADS;
    constexpr int SEG = EPT_MAX * T;
    uint32_t *cnt = smem;                     // [T]   records per output pixel
    uint32_t *wsum = smem + T;                // [T/64] wave sums of the scan
    uint16_t *off = reinterpret_cast<uint16_t *>(smem + T + 16);            // [T] exclusive prefix (< 2^16)
    constexpr bool R6 = rec6(EPT_MAX, SCAN);
    float *rec_w = reinterpret_cast<float *>(smem + T + 16 + T / 2);        // R6: [rec_cap] weights ...
    uint16_t *rec_e = reinterpret_cast<uint16_t *>(rec_w + rec_cap(EPT_MAX));   // ... and [rec_cap] entry indices (< SEG <= 2^16)
    uint2 *rec = reinterpret_cast<uint2 *>(smem + T + 16 + T / 2);          // !R6: [rec_cap] (entry index, weight bits)
    auto REC_PUT = [&](uint32_t i, uint32_t e, float w) {
        if constexpr (R6) { rec_e[i] = (uint16_t)e; rec_w[i] = w; } else rec[i] = make_uint2(e, __float_as_uint(w));
    };
    auto REC_E = [&](uint32_t i) -> uint32_t { if constexpr (R6) return rec_e[i]; else return rec[i].x; };
    auto REC_W = [&](uint32_t i) -> float { if constexpr (R6) return rec_w[i]; else return __uint_as_float(rec[i].y); };
    float4 *val4 = reinterpret_cast<float4 *>(reinterpret_cast<char *>(smem) + lds_head_bytes(EPT_MAX, SCAN) + SLR_LDS_PAD);     // [SEG][CHUNK/4] staged source values

    // Workgroup b runs on XCD b % 8 (observed dispatch order; speed only, never correctness).
    // Groups of XCD_GROUP consecutive work items (= horizontally neighbouring tiles) are placed on
    // the same XCD, groups round-robin over the XCDs: a tile's column halo (bin entries owned by
    // the tile to its left: one 128-byte line per row for 4 useful bytes) is then served by that
    // XCD's L2 instead of HBM, while heavy image regions still spread over all XCDs.
    uint32_t item;
    if (SCAN && !ROWS) {                               // no plan: block -> tile (same XCD grouping), one workgroup per tile
        const uint32_t slot = bx >> 3;
        item = ((slot / XCD_GROUP) * 8u + (bx & 7u)) * XCD_GROUP + slot % XCD_GROUP;
        if (item >= a.nt) return;
#ifdef SLR_SCAN_ORDER_HOOK
        if (a.order) item = a.order[bx < a.nt ? bx : 0];          // experiment: tiles in a given order (heaviest first)
#endif
    } else if (!WHOLE) {
        const uint32_t total = a.totals[0];
        const uint32_t slot = bx >> 3;
        item = ((slot / XCD_GROUP) * 8u + (bx & 7u)) * XCD_GROUP + slot % XCD_GROUP;
        if (item >= total) return;
    } else {                                           // the (rare) whole-tile items, grid-strided
        if (bx >= a.totals[4]) return;
        item = a.whole_items[bx];
    }
  // SCAN: the current piece of work -- the block's own tile first, then segments of shared tiles (w_piece)
  const int wave_in_group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t w_grp = blockIdx.y, w_seg = 0, w_po = 0;
  bool w_piece = false;
  for (uint32_t wi = bx;;) {                           // one pass unless WHOLE (its items) or SCAN (popped segments)
    ItemDesc it;
    uint32_t item_at = item;                       // ROWS: position of the item in items[]
    if (SCAN && !ROWS) { it = ItemDesc{}; it.tile = item; it.nseg = 1; }
    else if (ROWS && !WHOLE) { const uint32_t nh = a.totals[5]; item_at = item < nh ? item : a.items_cap - 1u - (item - nh); it = a.items[item_at]; }   // heavy from the front, the rest from the back
    else it = a.items[item];
    const uint32_t t = it.tile;
    // Normally one workgroup = one segment.  A tile whose segments did not fit into the partial-slot
    // budget (pathological flows: everything converging into a few tiles) is walked segment by
    // segment by ONE workgroup: every work-item keeps accumulating its own output pixel through
    // global memory (its own earlier store), and normalises after the last segment.
    // That is a separate instantiation (WHOLE) launched after the main one; the main kernel skips
    // those items, so its code carries no loop.
    constexpr bool whole = WHOLE || (SCAN && !ROWS);   // scan front end: a tile with more than SEG entries is walked pass by pass as well
    if (!SCAN && (it.nseg == 0) != WHOLE) return;
    // Channel groups (gridDim.y > 1: small grids, see launch_batch): this workgroup builds the tile's records like
    // any other and gathers the planes [cb, cend) only.  Groups start on a multiple of 2 * CHUNK planes.
    const int cper = (((a.C + (int)gridDim.y - 1) / (int)gridDim.y + 2 * CHUNK - 1) / (2 * CHUNK)) * (2 * CHUNK);
    const int cb = (int)w_grp * cper, cend = min(a.C, cb + cper);
    if (cb >= a.C) return;
    uint32_t nloop = (WHOLE && !ROWS) ? (it.cnt0 + it.cnt1 + (uint32_t)a.seg - 1) / (uint32_t)a.seg : 1u;   // (rows, WHOLE: set after its count pass)
    float nrm_total = 0.0f, g2_sum = 0.0f, g2_nrm = 0.0f;
    const int n = t / a.tiles, tl = t - n * a.tiles;
    const int ty0 = (tl / a.tiles_x) * TILE_H, tx0 = (tl % a.tiles_x) * TILE_W;
    const int HW = a.H * a.W;
    // SCAN: the body is a loop (claimed segments); everything derived from the work-item index would be hoisted out of it and
    // kept in registers across the whole body (+16 VGPRs: over the 128 of two workgroups per CU) -- so the index is rebuilt in
    // every round from the wave's number (a scalar) and the lane count, and made to look loop-variant
    uint32_t ones_ = ~0u;
    if (SCAN && !ROWS) asm volatile("" : "+s"(ones_));           // (a scalar the optimiser cannot see through)
    const int tid = (SCAN && !ROWS) ? wave_in_group * 64 + (int)__builtin_amdgcn_mbcnt_hi(ones_, __builtin_amdgcn_mbcnt_lo(ones_, 0u)) : (int)threadIdx.x;

    // ---------------- SCAN front end: the tile's entry list is built here, in LDS, from the flow itself
    // Candidates: source tiles whose destination box (scan_box_kernel) touches this tile -> a bit mask in LDS (order-free
    // atomicOr), walked in index order by every wave; wave w scans row w of each candidate (coalesced), CB candidates'
    // loads in flight together.  A hit (some corner of the pixel's footprint inside this tile) becomes an entry
    // (source pixel, flow) in LDS.  Mode 0 hands out entry slots with one LDS atomic per wave and candidate; if the tile
    // turns out to hold more than SEG entries, it is walked in passes of SEG entries whose membership must be the same
    // in every pass: mode 1 counts the hits per wave, mode 2 emits the entries with ordinals (wave, candidate, lane)
    // inside [lo, hi).
    uint32_t *cmask = reinterpret_cast<uint32_t *>(off);          // [64] candidate bits of a block of 2048 source tiles, [64] entry
                                                                  // counter, [65 + w] hits of wave w   (off[] is free until phase 1b)
    uint32_t *ent_pix = reinterpret_cast<uint32_t *>(val4);       // [SEG] entries of the current pass ...
    float *ent_fx = reinterpret_cast<float *>(val4) + SEG, *ent_fy = ent_fx + SEG;   // ... (val4 is free until the first chunk is staged)
    uint32_t wave_base = 0, scan_total = 0;
    uint32_t *clist = reinterpret_cast<uint32_t *>(smem + T + 16 + T / 2);   // [2048] candidate source tiles of the block, in index
                                                                             // order (the record area is free until phase 1c)
    auto scan = [&](auto mode_tag, uint32_t lo, uint32_t hi) -> uint32_t {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int CB = SLR_SCAN_CB;
        const int lane = tid & 63, wid = tid >> 6;
        const SrcBox *boxes = a.box + (size_t)n * a.tiles;
        const float *fl = a.flow[0] + (size_t)n * 2 * HW;
        const float inv_tx = 1.0f / (float)a.tiles_x;
        uint32_t wcount = 0;                                      // hits of this wave so far (modes 1, 2)
        for (int base = 0; base < a.tiles; base += 2048) {
            // the box loads do not depend on LDS: issue them first
            SrcBox bx4[2048 / T];
#pragma unroll
            for (int k = 0; k < 2048 / T; ++k) {
                const int st = base + tid + k * T;
                bx4[k] = boxes[st < a.tiles ? st : 0];
            }
            if (tid < 64) cmask[tid] = 0;
            if (MODE == 0 && tid == 64 && base == 0) cmask[64] = 0;
            __syncthreads();
            if (MODE == 0) SLR_STAMP(33);
#pragma unroll
            for (int k = 0; k < 2048 / T; ++k) {
                const int st = base + tid + k * T;
                const SrcBox b = bx4[k];
                if (st < a.tiles && b.x1 >= tx0 - 1 && b.x0 <= tx0 + TILE_W - 1 && b.y1 >= ty0 - 1 && b.y0 <= ty0 + TILE_H - 1)
                    atomicOr(&cmask[(st - base) >> 5], 1u << (st & 31));
            }
            __syncthreads();
            if (MODE == 0) SLR_STAMP(34);
            // Every wave expands the bit mask into the ordered candidate list itself (lane l owns word l; a wave scan places its
            // bits): the eight waves write identical values to the same LDS words, and each reads back only after its own writes.
            const uint32_t word = cmask[lane];
            const uint32_t pc = (uint32_t)__popc(word);
            uint32_t inc = pc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            const int nc = (int)__shfl(inc, 63);
#ifdef SLR_SCAN_STATS
            if (tid == 0 && a.ctl) { atomicAdd(&a.ctl[8 + MODE], 1u); atomicAdd(&a.ctl[12 + MODE], (uint32_t)nc); }
#endif
            {
                uint32_t w = word, at = inc - pc;
                while (w) {
                    const int bit = __ffs((int)w) - 1;
                    w &= w - 1;
                    clist[at++] = (uint32_t)(base + lane * 32 + bit);
                }
            }
            // CB candidates per group: one row of each (wave w = row w), all flow loads of a group in flight together, and the
            // next group's loads issued before this group's hits are processed
            // (a group keeps the flow values per lane and the candidates' tile coordinates as wave-uniform scalars: the pixel
            // coordinates are recomputed where they are needed -- the scan is where this kernel's register pressure peaks)
            struct Group { float fx[CB], fy[CB]; int stx[CB], sty[CB]; };
            auto issue = [&](Group &g, int k0) {
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const int k = k0 + i;
                    const int st = __builtin_amdgcn_readfirstlane((int)clist[k < nc ? k : 0]);   // (uniform address: one broadcast read)
                    const int sty = (int)(((float)(st) + 0.5f) * inv_tx);      // st / tiles_x, exact for st < 2^22
                    g.sty[i] = k < nc ? sty : -1;
                    g.stx[i] = st - sty * a.tiles_x;
                    const int sy = sty * TILE_H + wid, sx = g.stx[i] * TILE_W + lane;
                    const bool in = (k < nc) & (sy < a.H) & (sx < a.W);
                    const int q = in ? sy * a.W + sx : 0;
                    g.fx[i] = fl[q];
                    g.fy[i] = fl[HW + q];
                }
            };
            auto process = [&](const Group &g) {
                unsigned long long hm[CB];
                uint32_t pre[CB], tot = 0;
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const int sy = g.sty[i] * TILE_H + wid, sx = g.stx[i] * TILE_W + lane;
                    const bool in = (g.sty[i] >= 0) & (sy < a.H) & (sx < a.W);
                    const Corners c = make_corners(g.fx[i], g.fy[i], sx, sy);
                    const int lx = c.x0 - tx0, ly = c.y0 - ty0;
                    const bool xa = (lx >= 0) & (lx < TILE_W) & (c.x0 < a.W), xb = (lx + 1 >= 0) & (lx + 1 < TILE_W) & (c.x0 + 1 < a.W);
                    const bool ya = (ly >= 0) & (ly < TILE_H) & (c.y0 < a.H), yb = (ly + 1 >= 0) & (ly + 1 < TILE_H) & (c.y0 + 1 < a.H);
                    hm[i] = __ballot(in & c.ok & (xa | xb) & (ya | yb));
                    pre[i] = tot;
                    tot += (uint32_t)__popcll(hm[i]);
                }
                if (tot == 0) return;                             // wave-uniform
                uint32_t b0;
                if (MODE == 0) {                                  // ONE slot reservation per wave and group
                    b0 = 0;
                    if (lane == 0) b0 = atomicAdd(&cmask[64], tot);
                    b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
                } else {
                    b0 = wave_base + wcount;
                    wcount += tot;
                }
                if (MODE != 1) {
#pragma unroll
                    for (int i = 0; i < CB; ++i) {
                        const uint32_t slot = b0 + pre[i] + (uint32_t)__popcll(hm[i] & ((1ull << lane) - 1ull));
                        if (((hm[i] >> lane) & 1ull) && slot >= lo && slot < hi) {
                            ent_pix[slot - lo] = (uint32_t)((g.sty[i] * TILE_H + wid) * a.W + g.stx[i] * TILE_W + lane);
                            ent_fx[slot - lo] = g.fx[i];
                            ent_fy[slot - lo] = g.fy[i];
                        }
                    }
                }
            };
            Group ga, gb;
            if (nc > 0) issue(ga, 0);
            for (int k0 = 0; k0 < nc; k0 += 2 * CB) {
                if (k0 + CB < nc) issue(gb, k0 + CB);
                process(ga);
                if (k0 + CB < nc) {
                    if (k0 + 2 * CB < nc) issue(ga, k0 + 2 * CB);
                    process(gb);
                }
            }
            if (MODE == 0) SLR_STAMP(35);
            __syncthreads();
        }
        return MODE == 0 ? cmask[64] : wcount;
    };
    // ROWS: the entries of this piece of work from the tile's row-segment list.  A piece = a range of the tile's output columns
    // (whole column octants, cut by the plan): it owns its output pixels, so nothing is summed across pieces -- no partial
    // tiles, no combine.  Entry slots:
    //   mode 0  (whole tile, <= SEG entries): the list's hit counts give every row segment its first slot (exclusive scan);
    //   mode 1  (a column range, or a tile whose list overflowed and that walks ALL row segments of its sample): one LDS atomic
    //           per wave and row segment; more than SEG hits -> the piece is handed to the pass-by-pass launch (WHOLE);
    //   mode 2  (WHOLE): ordinals (wave w takes segments w, w + 8, ...: hits of the waves before + own so far) after a count
    //           pass; pass si emits the ordinals [si * SEG, (si + 1) * SEG).
    const int p_oct = ROWS ? (int)min(max(it.nseg, 1u), 8u) : 8;             // the piece: p_oct column octants from octant it.seg
    const int pw = 8 * p_oct;                                                // its width in output columns (8 .. 64)
    const int pca = ROWS ? 8 * (int)it.seg : 0, pcb = pca + pw;              // its output columns [pca, pcb)
    // A narrow piece keeps the tile's shape -- wave w = output row w -- and gives every output pixel G = 8, 4 or 2 lanes (8, 16,
    // <= 32 columns): a ridge piles ~750 entries onto 64 pixels, lists of 150 records that one lane per pixel walked for
    // 170-250 us while the rest of its wave sat idle; G lanes walk a list strided and add up through log2(G) cross-lane steps.
    const int g_log = !(ROWS && SLR_ROWS_GROUP) ? 0 : pw <= 8 ? 3 : pw <= 16 ? 2 : (pw <= 32 && SLR_ROWS_GROUP > 1) ? 1 : 0;
    const int pid = ROWS ? (tid & ~63) | ((tid & 63) >> g_log) : tid;       // the output pixel (tile-local index) this work-item serves
    const bool rows_ovf = ROWS && it.cnt1 > (uint32_t)ROW_CAP;
    uint32_t rows_n = 0;                                           // row segments to walk
    auto rows_setup = [&]() {                                      // list -> LDS (image order) + first slots; needs the record area
        const int lane = tid & 63, wid = tid >> 6;
        uint32_t *rl_sy = clist, *rl_sx = clist + ROW_CAP, *rl_base = clist + 2 * ROW_CAP;      // [ROW_CAP], [ROW_CAP], [ROW_CAP + 1]
        rows_n = rows_ovf ? (uint32_t)a.H * (uint32_t)a.tiles_x : it.cnt1;
        if (tid == 0) cmask[64] = 0;
        if (!rows_ovf) {
            RowRec r = {0u, 0u};
            if ((uint32_t)tid < rows_n) r = a.rowlist[(size_t)t * ROW_CAP + tid];
            uint32_t c = r.sx_cnt & 0xffu;
#if SLR_ROW_SORT
            // the appends arrived in any order: put the list into image order (row, column) -- the order the bins have, which the
            // staging loads and the record lists like best -- by ranking every key among the others (<= ROW_CAP broadcast reads)
            const unsigned long long mykey = ((unsigned long long)(r.sy & 0xffffffu) << 24) | (r.sx_cnt >> 8);
            if ((uint32_t)tid < rows_n) { rl_sy[tid] = r.sy; rl_sx[tid] = r.sx_cnt >> 8; }
            __syncthreads();
            uint32_t rank = 0;
            if ((uint32_t)tid < rows_n)
                for (uint32_t q = 0; q < rows_n; ++q) {
                    const unsigned long long k2 = ((unsigned long long)(rl_sy[q] & 0xffffffu) << 24) | rl_sx[q];
                    rank += (k2 < mykey) ? 1u : 0u;                    // (keys are distinct: one append per (segment, tile))
                }
            __syncthreads();
            if ((uint32_t)tid < rows_n) { rl_sy[rank] = r.sy; rl_sx[rank] = r.sx_cnt >> 8; rl_base[rank] = c; }
            __syncthreads();
            c = (uint32_t)tid < rows_n ? rl_base[tid] : 0u;            // counts in sorted order
            __syncthreads();
#else
            if (tid < ROW_CAP) { rl_sy[tid] = r.sy; rl_sx[tid] = r.sx_cnt >> 8; }
#endif
            uint32_t inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            if (lane == 63) wsum[wid] = inc;
            __syncthreads();
            uint32_t woff = 0;
#pragma unroll
            for (int w = 0; w < T / 64; ++w) woff += (w < wid) ? wsum[w] : 0u;
            if (tid <= ROW_CAP) rl_base[tid] = woff + inc - c;
        }
        __syncthreads();
    };
    uint32_t rows_wave_base = 0;
    // one walk over this wave's row segments; returns the wave's hits.  MODE as above; EMIT: write the entries with slots in [lo, hi)
    auto rows_walk = [&](auto mode_tag, auto emit_tag, uint32_t lo, uint32_t hi) -> uint32_t {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool EMIT = decltype(emit_tag)::value;
        constexpr int CB = SLR_ROW_CB;
        const int lane = tid & 63, wid = tid >> 6;
        const float *fl = a.flow[0] + (size_t)n * 2 * HW;
        const uint32_t *rl_sy = clist, *rl_sx = clist + ROW_CAP, *rl_base = clist + 2 * ROW_CAP;
        const uint32_t my_n = rows_n > (uint32_t)wid ? (rows_n - (uint32_t)wid + (uint32_t)(T / 64) - 1u) / (uint32_t)(T / 64) : 0u;
        const uint32_t range_mask = ((1u << p_oct) - 1u) << (pca >> 3);            // the piece's column octants
        struct Group { float fx[CB], fy[CB]; int sy[CB], stx[CB]; uint32_t b0[CB]; };
        auto issue = [&](Group &g, uint32_t j0) {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const uint32_t j = j0 + (uint32_t)i, ri = (uint32_t)wid + j * (uint32_t)(T / 64);
                bool on = j < my_n;
                int sy, stx;
                uint32_t base = 0;
                if (rows_ovf) {
                    sy = (int)(ri / (uint32_t)a.tiles_x);
                    stx = (int)(ri - (uint32_t)sy * (uint32_t)a.tiles_x);
                } else {
                    const uint32_t q = on ? ri : 0u;
                    const uint32_t syw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl_sy[q]);
                    sy = (int)(syw & 0xffffffu);
                    on = on && ((syw >> 24) & range_mask) != 0u;       // (a piece only loads the segments that touch its column octants)
                    stx = __builtin_amdgcn_readfirstlane((int)rl_sx[q]);
                    if (MODE == 0) base = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl_base[q]);
                }
                g.sy[i] = on ? sy : -1;
                g.stx[i] = stx;
                g.b0[i] = base;
                const int sx = stx * TILE_W + lane;
                const bool in = on & (sx < a.W);
                const int q = in ? sy * a.W + sx : 0;
                g.fx[i] = fl[q];
                g.fy[i] = fl[HW + q];
            }
        };
        uint32_t wcount = 0;
        auto process = [&](const Group &g) {
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int sy = g.sy[i], sx = g.stx[i] * TILE_W + lane;
                const bool in = (sy >= 0) & (sx < a.W);
                const Corners c = make_corners(g.fx[i], g.fy[i], sx, sy);
                const int lx = c.x0 - tx0, ly = c.y0 - ty0;
                const bool xa = (lx >= pca) & (lx < pcb) & (c.x0 < a.W), xb = (lx + 1 >= pca) & (lx + 1 < pcb) & (c.x0 + 1 < a.W);
                const bool ya = (ly >= 0) & (ly < TILE_H) & (c.y0 < a.H), yb = (ly + 1 >= 0) & (ly + 1 < TILE_H) & (c.y0 + 1 < a.H);
                const bool hit = in & c.ok & (xa | xb) & (ya | yb);
                const unsigned long long hm = __ballot(hit);
                const uint32_t pc = (uint32_t)__popcll(hm);
                if (sy < 0) continue;                                  // (wave-uniform: no row segment here)
                uint32_t b0;
                if (MODE == 0) b0 = g.b0[i];
                else if (MODE == 1) {
                    b0 = 0;
                    if (pc) { if (lane == 0) b0 = atomicAdd(&cmask[64], pc); b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0); }
                } else b0 = rows_wave_base + wcount;
                wcount += pc;
                if (EMIT) {
                    const uint32_t slot = b0 + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                    if (hit && slot >= lo && slot < hi) {
                        ent_pix[slot - lo] = (uint32_t)(sy * a.W + sx);
                        ent_fx[slot - lo] = g.fx[i];
                        ent_fy[slot - lo] = g.fy[i];
                    }
                }
            }
        };
        Group ga, gb;
        if (my_n > 0) issue(ga, 0u);
        for (uint32_t j0 = 0; j0 < my_n; j0 += 2 * CB) {
            if (j0 + CB < my_n) issue(gb, j0 + CB);
            process(ga);
            if (j0 + CB < my_n) {
                if (j0 + 2 * CB < my_n) issue(ga, j0 + 2 * CB);
                process(gb);
            }
        }
        return wcount;
    };
    if (!w_piece) { SLR_STAMP(41); SLR_STAMP_RT(48); }
    int qavail = 0;                                               // the semaphore as seen two chunks before the end of this work
    // (a grid that fits the chip in one round ends all at once with nothing else to do: everybody helps)
    const bool helper = ((bx >> 3) % (uint32_t)SLR_SHARE_HELPERS) == 0u || gridDim.x * gridDim.y <= (uint32_t)SLR_CSPLIT_SLOTS;
    bool part = false;                                            // this piece of work is ONE segment of a shared tile
    bool ctx = false;                                             // this workgroup holds a shared tile it may draw more tickets of
    uint32_t ns_tile = 1;                                         // segments of the tile (part)
    if (ROWS && !WHOLE) {
        rows_setup();
        if (it.nseg >= 8u && it.cnt0 <= (uint32_t)SEG && !rows_ovf) {
            rows_walk(std::integral_constant<int, 0>{}, std::true_type{}, 0u, (uint32_t)SEG);
            scan_total = it.cnt0;                                 // exact (rowbin_kernel), <= SEG (the plan)
            __syncthreads();
        } else {
            rows_walk(std::integral_constant<int, 1>{}, std::true_type{}, 0u, (uint32_t)SEG);
            __syncthreads();
            scan_total = cmask[64];
            if (scan_total > (uint32_t)SEG) {                     // (uniform) more than one pass: the pass-by-pass launch takes it
                // (one entry per piece: every channel group of the piece gets here, the pass-by-pass workgroup covers all planes)
                if (tid == 0 && blockIdx.y == 0) const_cast<uint32_t *>(a.whole_items)[atomicAdd(const_cast<uint32_t *>(a.totals) + 4, 1u)] = item_at;
                return;
            }
        }
        SLR_CUT_AT(1);
    } else if (ROWS) {                                            // WHOLE: count, then nloop passes over the ordinals
        rows_setup();
        const uint32_t wc = rows_walk(std::integral_constant<int, 2>{}, std::false_type{}, 0u, 0u);
        if ((tid & 63) == 0) cmask[65 + (tid >> 6)] = wc;
        __syncthreads();
        uint32_t all = 0, wb = 0;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) { const uint32_t c = cmask[65 + w]; all += c; wb += w < (tid >> 6) ? c : 0u; }
        scan_total = all;
        rows_wave_base = wb;
        nloop = max(1u, (all + (uint32_t)SEG - 1u) / (uint32_t)SEG);
        __syncthreads();
    } else if (SCAN) {
        if (!w_piece) scan_total = scan(std::integral_constant<int, 0>{}, 0u, (uint32_t)SEG);
        if (!w_piece) SLR_STAMP(42);
        SLR_CUT_AT(1);
        if (w_piece || scan_total > (uint32_t)SEG) {              // heavy tile: segments with reproducible membership
            {
                const uint32_t wc = scan(std::integral_constant<int, 1>{}, 0u, 0u);
                if ((tid & 63) == 0) cmask[65 + (tid >> 6)] = wc;
                __syncthreads();
                uint32_t all = 0, wb = 0;
#pragma unroll
                for (int w = 0; w < T / 64; ++w) { const uint32_t c = cmask[65 + w]; all += c; wb += w < (tid >> 6) ? c : 0u; }
                scan_total = all;                                 // (equals the optimistic count; a claimed segment has no other)
                wave_base = wb;
                __syncthreads();
            }
            ns_tile = (scan_total + SEG - 1) / SEG;
            part = ctx = w_piece;
            nloop = part ? 1u : ns_tile;                          // not shared: this workgroup walks all passes itself
            if (!w_piece && a.ctl && ns_tile <= SHARE_MAX_SEG && t <= SHARE_MAX_TILE) {
                // reserve ns slots; if they exist, push segments 1..ns-1 and keep segment 0
                if (tid == 0) {
                    const uint32_t po = atomicAdd(&a.ctl[2], ns_tile);
                    uint32_t qp = 0xffffffffu;
                    if (po + ns_tile <= a.part_slots && po + ns_tile <= SHARE_MAX_SLOT) qp = atomicAdd(&a.ctl[1], ns_tile - 1u);
                    wsum[0] = po; wsum[1] = qp;
                }
                __syncthreads();
                const uint32_t po = (uint32_t)__builtin_amdgcn_readfirstlane((int)wsum[0]), qp = (uint32_t)__builtin_amdgcn_readfirstlane((int)wsum[1]);
                __syncthreads();
                if (qp != 0xffffffffu) {                          // (sum of ns - 1 over successful reservations < part_slots = queue size)
                    for (uint32_t i = tid; i + 1 < ns_tile; i += T)
                        __hip_atomic_store((gu64 *)(a.q_items + qp + i), pack_work(t, i + 1u, ns_tile, po, w_grp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the words are in memory before the semaphore says so
                    __syncthreads();
                    if (tid == 0) atomicAdd(&a.ctl[3], ns_tile - 1u);
                    part = ctx = true; nloop = 1; w_seg = 0; w_po = po;
                }
#ifdef SLR_SCAN_STATS
                if (tid == 0) { atomicAdd(&a.ctl[16], 1u); atomicAdd(&a.ctl[17], ns_tile); atomicMax(&a.ctl[18], ns_tile); if (!ctx) atomicAdd(&a.ctl[19], 1u); }
#endif
            }
        }
    }

  for (uint32_t si = 0; si < nloop; ++si) {
    const uint32_t s = ROWS ? (WHOLE ? si : 0u) : (SCAN && part) ? w_seg : whole ? si : it.seg;
    const bool first = si == 0, last = si + 1 == nloop;
    if (!w_piece && first) SLR_STAMP(0);
    if (!w_piece && first && SCAN && (nloop > 1 || part)) SLR_STAMP(36);
    if (ROWS && WHOLE) {
        if (si > 0) rows_setup();                                 // (the list shares LDS with the previous pass's records)
        rows_walk(std::integral_constant<int, 2>{}, std::true_type{}, s * (uint32_t)SEG, (s + 1) * (uint32_t)SEG);
        __syncthreads();
    } else if (SCAN && !ROWS && (nloop > 1 || part)) scan(std::integral_constant<int, 2>{}, s * (uint32_t)SEG, (s + 1) * (uint32_t)SEG);
    cnt[tid] = 0;
    __syncthreads();

    // ---------------- phase 1a: footprints of this work-item's bin entries, slot reservation
    constexpr bool G2OK = NORM && !MAXOP && !SCAN && CHUNK == 4 && EPT_MAX == EPT_TWO;
    const bool g2 = G2OK && a.in2 != nullptr;
    const float shift = (a.mulmode == MUL_EXP_SHIFT) ? a.mulmax[0] : 0.0f;
    const bool has_mul = a.mulmode != MUL_ONE;
    const uint32_t lo = s * (uint32_t)a.seg, hi = lo + (uint32_t)a.seg;   // range of the concatenated bin
    uint32_t e_pix[EPT_MAX];
    float e_m[EPT_MAX];              // g2: the first group's weight of the entry (applied when its values are staged)
    bool e_val[EPT_MAX];
    uint32_t e_ts[EPT_MAX][4];      // (output pixel << 16) | slot, 0xffffffff = corner not in tile
    float e_w[EPT_MAX][4];
#pragma unroll
    for (int j = 0; j < EPT_MAX; ++j) {
        e_pix[j] = 0;
        e_m[j] = 1.0f;
        if (g2) val4[vslot<CHUNK>(tid + j * T, 0)] = make_float4(0.f, 0.f, 0.f, 0.f);     // (entries past the bin: no records point here)
#pragma unroll
        for (int k = 0; k < 4; ++k) { e_ts[j][k] = 0xffffffffu; e_w[j][k] = 0.0f; }
    }
    const float *ip = a.in + (size_t)n * a.C * HW;
    const int cmax = cend - 1;                       // (prefetches past the group's last plane re-read it)
    auto prefetch = [&](float (&pre)[EPT_MAX][CHUNK], int c0) {
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const float *plane = ip + (size_t)min(c0 + u, cmax) * HW;
#pragma unroll
            for (int j = 0; j < EPT_MAX; ++j) pre[j][u] = plane[e_pix[j]];
        }
    };
    float preA[EPT_MAX][CHUNK], preB[EPT_MAX][CHUNK];
    {
        // entry j of this work-item = element lo + tid + j*T of [bin(flow0) ; bin(flow1)]
        const uint32_t c0 = it.cnt0, c1 = it.cnt1;
        const uint32_t *l0 = SCAN ? nullptr : a.list[0] + it.off0;
        const uint32_t *l1 = SCAN ? nullptr : a.ndir > 1 ? a.list[1] + it.off1 : l0;
        const float *mp = has_mul ? a.mul + (size_t)n * HW : nullptr;
        bool (&val)[EPT_MAX] = e_val;
        int dir[EPT_MAX];
        // independent index loads first, then the dependent flow / weight loads
        float fx[EPT_MAX], fy[EPT_MAX], mm[EPT_MAX];
#pragma unroll
        for (int j = 0; j < EPT_MAX; ++j) {
            if (SCAN) {                                            // entries of this pass: in LDS, with their flow
                const uint32_t k = tid + j * T;
                val[j] = lo + k < scan_total && k < (uint32_t)SEG;
                dir[j] = 0;
                e_pix[j] = val[j] ? ent_pix[k] : 0u;
                fx[j] = val[j] ? ent_fx[k] : 0.0f;
                fy[j] = val[j] ? ent_fy[k] : 0.0f;
            } else {
                const uint32_t k = lo + tid + j * T;
                val[j] = (k < hi) & (k < c0 + c1);
                dir[j] = (k >= c0) ? 1 : 0;
                e_pix[j] = val[j] ? (dir[j] ? l1[k - c0] : l0[k]) : 0u;
            }
        }
        SLR_STAMP(28);
        // the plane loads of the first two chunks only need the source indices: issue them now,
        // they complete under the footprint math, the LDS atomics and the scan
        prefetch(preA, cb);
        prefetch(preB, cb + CHUNK);
#if SLR_DBG & 8
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (measurement: raw latency of the first plane loads)
        SLR_STAMP(29);
#endif
#pragma unroll
        for (int j = 0; j < EPT_MAX; ++j) {
            if (!SCAN) {
                const float *fl = a.flow[dir[j]] + (size_t)n * 2 * HW;
                fx[j] = val[j] ? fl[e_pix[j]] : 0.0f;
                fy[j] = val[j] ? fl[HW + e_pix[j]] : 0.0f;
            }
            mm[j] = (val[j] && has_mul) ? mp[e_pix[j]] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < EPT_MAX; ++j) {
            if (!val[j]) continue;
            const uint32_t pix = e_pix[j];
            const int y = pix / a.W, x = pix - y * a.W;
            const Corners c = make_corners(fx[j], fy[j], x, y);
            float m = a.scale[dir[j]];
            if (a.mulmode == MUL_PLANE) m = mm[j] * m;
            else if (a.mulmode >= MUL_EXP) m = expf(mm[j] - shift) * m;
            if (g2) {
                // two weight groups share the records: the records keep the PURE bilinear weights, the first group's weight
                // m multiplies its values when they are staged, and the entry's slot of the special chunk (gathered
                // before the value planes) carries  m | in2 * m2 | m2  -> the two normalisers and the second group's sum
                float m2 = a.scale[dir[j]];
                const float l2 = a.mul2[(size_t)n * HW + pix];
                m2 = a.mulmode2 == MUL_PLANE ? l2 * m2 : expf(l2) * m2;
                val4[vslot<CHUNK>(tid + j * T, 0)] = make_float4(m, a.in2[(size_t)n * HW + pix] * m2, m2, 0.0f);
                e_m[j] = m;
                m = 1.0f;
            }
            const int lx = c.x0 - tx0, ly = c.y0 - ty0;
            const bool xa = c.ok & (lx >= pca) & (lx < pcb) & (c.x0 < a.W);          // (rows front end: this piece's output columns only)
            const bool xb = c.ok & (lx + 1 >= pca) & (lx + 1 < pcb) & (c.x0 + 1 < a.W);
            const bool ya = (ly >= 0) & (ly < TILE_H) & (c.y0 < a.H);
            const bool yb = (ly + 1 >= 0) & (ly + 1 < TILE_H) & (c.y0 + 1 < a.H);
            const int oc = ly * TILE_W + lx - pca;        // tile-local output pixel (a piece's columns start at lane 0 of the row's wave)
            const bool kb[4] = {bool(xa & ya), bool(xb & ya), bool(xa & yb), bool(xb & yb)};
            const int tg[4] = {oc, oc + 1, oc + TILE_W, oc + TILE_W + 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (kb[k]) {
                    const uint32_t slot = atomicAdd(&cnt[tg[k]], 1u);          // ds_add_rtn_u32
                    e_ts[j][k] = ((uint32_t)tg[k] << 16) | slot;
                    e_w[j][k] = has_mul || a.ndir > 1 ? m * c.w[k] : c.w[k];
                }
            }
        }
    }
    SLR_STAMP(1);
    __syncthreads();
    SLR_STAMP(2);
    SLR_CUT_AT(2);

    // ---------------- phase 1b: exclusive scan of the counts (T values, one per work-item)
    {
        const int lane = tid & 63, wid = tid >> 6;
        const uint32_t v = cnt[tid] | 1u;              // odd list length (bank spreading, see rec_cap)
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) woff += (w < wid) ? wsum[w] : 0u;
        off[tid] = (uint16_t)(woff + inc - v);
    }
    __syncthreads();

    // ---------------- phase 1c: scatter the records into the per-output-pixel lists
#pragma unroll
    for (int j = 0; j < EPT_MAX; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ts = e_ts[j][k];
            if (ts != 0xffffffffu) {
                REC_PUT(off[ts >> 16] + (ts & 0xffffu), tid + j * T, e_w[j][k]);
            }
        }
    __syncthreads();

    SLR_STAMP(3);
    SLR_CUT_AT(3);
    // ---------------- phase 2: stage a chunk of planes in LDS, gather per output pixel
    // A work-item walks at most LMAX records of its own list; what is left of a longer list (a
    // "sink" pixel where hundreds of sources converge) is walked by the whole wave, lane-strided,
    // and wave-reduced -- otherwise one lane serialises thousands of records in every chunk.
    // (rows front end, narrow pieces: lane g of a pixel's G lanes takes records g, g + G, ... of its list -- r0 is ITS first record;
    // everywhere else G = 1)
    const uint32_t r1 = off[pid] + cnt[pid];
    const uint32_t r0 = off[pid] + (ROWS ? (uint32_t)tid & ((1u << g_log) - 1u) : 0u);
    const int lane = tid & 63;
    uint32_t wave_recs = r1 - r0;                          // records of this wave's 64 output pixels
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_recs += __shfl_xor(wave_recs, d);
    // own share: twice the wave's average list length (a uniformly compressed region stays
    // per-lane), at least LMAX; what is left of a longer list goes to the whole wave once it exceeds
    // SLR_HEAVY_SLACK records (a cooperative pass costs ~50 cross-lane operations per chunk, a lane
    // walking alone ~10 per record while the other 63 wait: measured optimum 16-32 on the Euler clips)
#ifdef SLR_TRACE
    {
        __shared__ uint32_t dbg_max;
        if (tid == 0) dbg_max = 0;
        __syncthreads();
        atomicMax(&dbg_max, r1 - r0);
        __syncthreads();
        if (a.trace && tid == 0) {
            long long *tr = a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLR_TRACE_SLOTS;
            tr[43] = dbg_max; tr[44] = wave_recs; tr[45] = SCAN ? scan_total : it.cnt0 + it.cnt1; tr[46] = s;
        }
    }
#endif
    const uint32_t own = max((uint32_t)LMAX, 2u * ((wave_recs + 63u) >> 6));
    uint32_t rl = (r1 - r0 >= own + (uint32_t)SLR_HEAVY_SLACK) ? r0 + own : r1;
    unsigned long long heavy = __ballot(r1 > rl);
    // (the cooperative passes run one after the other, ~1.5 k cycles each per chunk, while lanes walking their own lists run side by
    // side: with many long lists in one wave -- a piece of a ridge tile: 750 entries into 64 pixels, 15 lists over the limit --
    // they cost 250 us per workgroup; more than SLR_HEAVY_MAX of them and every lane walks its own)
    if (__popcll(heavy) > SLR_HEAVY_MAX || (ROWS && g_log)) { rl = r1; heavy = 0ull; }
    const int ly = pid / TILE_W, lx = pca + pid - ly * TILE_W;
    const int oy = ty0 + ly, ox = tx0 + lx;
    const bool inside = (oy < a.H) & (ox < a.W) & (lx < pcb) & ((tid & ((1 << g_log) - 1)) == 0);   // (rows front end: the piece's
                                                                                    // columns; the first lane of a pixel's group stores)
    const bool single = ROWS || (it.nseg <= 1 && !(SCAN && part));   // results go straight to the output tensor
    float *op = single ? a.out + (size_t)n * a.C * HW + (size_t)oy * a.W + ox
                       : a.partial + (size_t)((SCAN ? w_po : it.partoff) + s) * a.part_stride + tid;
    const uint32_t ostride = single ? (uint32_t)HW : (uint32_t)TILE_PIX;      // (elements; HW < 2^29)
    // SCAN, shared tile: slot (w_po + segment), laid out [chunk of 4 planes][work-item][4] so that a chunk is ONE 16-byte
    // write-through store per work-item; the normaliser plane follows the last chunk
    float *const pslot = (SCAN && part) ? a.partial + (size_t)(w_po + s) * a.part_stride : nullptr;
    const size_t pnorm = (size_t)((a.C + 3) / 4) * 4 * TILE_PIX;
    // register-resident head of this pixel's record list (see the gather loop)
    constexpr uint32_t NULL_E = SEG;                   // staged-entry index of the all-zero slot
    constexpr int KREG = ROWS ? SLR_KREG_ROWS : SCAN ? SLR_KREG_SCAN : EPT_MAX == EPT_ONE ? SLR_KREG_ONE : SLR_KREG_TWO;   // 4 records for one flow, 6 for two (8 / 10: < 1 % gain)
    float cw[KREG];
    uint32_t ce[KREG];
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
        const bool on = r0 + ((uint32_t)k << g_log) < rl;
        const uint32_t qi = on ? r0 + ((uint32_t)k << g_log) : 0u;
        ce[k] = on ? REC_E(qi) : NULL_E;
        cw[k] = on ? REC_W(qi) : 0.0f;
    }
#pragma unroll
    for (int h = 0; h < CHUNK / 4; ++h)
        if (tid == 0) val4[vslot<CHUNK>(NULL_E, h)] = make_float4(0.f, 0.f, 0.f, 0.f);

    float nrm = 0.0f;
    if (NORM && !g2) {
        if (ROWS && g_log) {                               // G lanes per pixel: strided, then added up
            const uint32_t G = 1u << g_log;
            for (uint32_t r = r0; r < r1; r += G) nrm += REC_W(r);
            if (g_log >= 3) nrm += __shfl_xor(nrm, 4);
            if (g_log >= 2) nrm += __shfl_xor(nrm, 2);
            nrm += __shfl_xor(nrm, 1);
        } else
        for (uint32_t r = r0; r < rl; ++r) nrm += REC_W(r);
        for (unsigned long long hv = heavy; hv; hv &= hv - 1) {        // long lists: the wave walks them together
            const int src = __ffsll((long long)hv) - 1;
            const uint32_t hb = __shfl(rl, src), he = __shfl(r1, src);
            float part = 0.0f;
            for (uint32_t r = hb + lane; r < he; r += 64) part += REC_W(r);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
            if (lane == src) nrm += part;
        }
        nrm_total += nrm;
        nrm = nrm_total;                                  // all segments seen so far (whole-tile items)
        if (SCAN && part) {
            store_wt4(pslot + pnorm + tid, nrm);
        } else if (single) {
            if (a.norm_out && inside && last && cb == 0) a.norm_out[(size_t)n * HW + (size_t)oy * a.W + ox] = norm_value(nrm, a.norm_mode, a.eps);
        } else if (cb == 0) {
            a.partial[(size_t)(it.partoff + s) * a.part_stride + (size_t)a.C * TILE_PIX + tid] = nrm;
        }
    }

    // Chunk pipeline, prefetch depth 2: while chunk c is gathered from LDS, the plane loads of
    // chunks c+1 and c+2 are in flight (registers preA / preB).  All global loads and stores are
    // UNCONDITIONAL -- invalid entries read pixel 0, planes past C re-read plane C-1, work-items
    // outside the image and planes past C store into the workspace's trash tile -- so the
    // compiler knows how many memory operations are younger than the ones it has to wait for
    // and emits s_waitcnt vmcnt(N) with N > 0 instead of draining everything.
    float *const trash = a.trash + tid;
    if (!inside && single) op = trash;
    const uint32_t ostr = (!inside && single) ? 0u : ostride;
    auto gather = [&](float (&acc)[CHUNK]) {
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) acc[u] = MAXOP ? a.init : 0.0f;
        // The first KREG records of the pixel live in registers (loaded once, reused by every chunk);
        // longer lists continue from LDS in batches of RB.  All LDS reads of a batch are issued
        // before its first FMA.  Missing records point at the zeroed NULL entry with weight 0:
        // fma(0, 0, acc) == acc, so the loop body needs no selects.
        {
            float v[KREG][CHUNK];
#pragma unroll
            for (int k = 0; k < KREG; ++k)
#pragma unroll
                for (int h = 0; h < CHUNK / 4; ++h) {
                    const float4 q = val4[vslot<CHUNK>(ce[k], h)];     // ds_read_b128: 4 planes per LDS instruction
                    v[k][4 * h] = q.x; v[k][4 * h + 1] = q.y; v[k][4 * h + 2] = q.z; v[k][4 * h + 3] = q.w;
                }
#pragma unroll
            for (int k = 0; k < KREG; ++k)
#pragma unroll
                for (int u = 0; u < CHUNK; ++u) {
                    if (MAXOP) acc[u] = fmaxf(ce[k] != NULL_E ? v[k][u] * cw[k] : -INFINITY, acc[u]);
                    else acc[u] = __builtin_fmaf(v[k][u], cw[k], acc[u]);
                }
        }
        for (uint32_t r = r0 + ((uint32_t)KREG << g_log); r < rl; r += (uint32_t)RB << g_log) {
            float w[RB];
            uint32_t e[RB];
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const bool on = r + ((uint32_t)k << g_log) < rl;
                const uint32_t qi = on ? r + ((uint32_t)k << g_log) : r;
                e[k] = on ? REC_E(qi) : NULL_E;
                w[k] = on ? REC_W(qi) : 0.0f;
            }
            float v[RB][CHUNK];
#pragma unroll
            for (int k = 0; k < RB; ++k)
#pragma unroll
                for (int h = 0; h < CHUNK / 4; ++h) {
                    const float4 q = val4[vslot<CHUNK>(e[k], h)];
                    v[k][4 * h] = q.x; v[k][4 * h + 1] = q.y; v[k][4 * h + 2] = q.z; v[k][4 * h + 3] = q.w;
                }
#pragma unroll
            for (int k = 0; k < RB; ++k)
#pragma unroll
                for (int u = 0; u < CHUNK; ++u) {
                    if (MAXOP) acc[u] = fmaxf(e[k] != NULL_E ? v[k][u] * w[k] : -INFINITY, acc[u]);
                    else acc[u] = __builtin_fmaf(v[k][u], w[k], acc[u]);
                }
        }
        for (unsigned long long hv = heavy; hv; hv &= hv - 1) {        // long lists, cooperatively
            const int src = __ffsll((long long)hv) - 1;
            const uint32_t hb = __shfl(rl, src), he = __shfl(r1, src);
            float part[CHUNK];
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) part[u] = MAXOP ? -INFINITY : 0.0f;
            for (uint32_t r = hb + lane; r < he; r += 64) {
                const uint32_t qe = REC_E(r);
                const float w = REC_W(r);
#pragma unroll
                for (int h = 0; h < CHUNK / 4; ++h) {
                    const float4 x = val4[vslot<CHUNK>(qe, h)];
                    const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        part[4 * h + i] = MAXOP ? fmaxf(xv[i] * w, part[4 * h + i]) : __builtin_fmaf(xv[i], w, part[4 * h + i]);
                }
            }
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    const float o = __shfl_xor(part[u], d);
                    part[u] = MAXOP ? fmaxf(part[u], o) : part[u] + o;
                }
                if (lane == src) acc[u] = MAXOP ? fmaxf(acc[u], part[u]) : acc[u] + part[u];
            }
        }
        if (ROWS && g_log) {                               // the G lanes of a pixel add up (every one of them ends up with the sum)
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                if (g_log >= 3) { const float o = __shfl_xor(acc[u], 4); acc[u] = MAXOP ? fmaxf(acc[u], o) : acc[u] + o; }
                if (g_log >= 2) { const float o = __shfl_xor(acc[u], 2); acc[u] = MAXOP ? fmaxf(acc[u], o) : acc[u] + o; }
                { const float o = __shfl_xor(acc[u], 1); acc[u] = MAXOP ? fmaxf(acc[u], o) : acc[u] + o; }
            }
        }
    };
    // ---- second weight group (g2): the special chunk staged in phase 1a, gathered with the same records BEFORE the value planes
    // acc2[0] = sum w*m (normaliser of the first group), acc2[1] = sum w*m2*in2, acc2[2] = sum w*m2
    if (g2) {
        __syncthreads();                               // the special values of all entries + the NULL slot are in LDS
        float acc2[CHUNK];
        gather(acc2);
        nrm_total += acc2[0];
        g2_sum += acc2[1];
        g2_nrm += acc2[2];
        nrm = nrm_total;                               // (all segments seen so far: whole-tile items)
        if (single) {
            if (last && cb == 0 && inside) {
                if (a.norm_out) a.norm_out[(size_t)n * HW + (size_t)oy * a.W + ox] = norm_value(nrm, a.norm_mode, a.eps);
                a.out2[(size_t)n * HW + (size_t)oy * a.W + ox] = finish(g2_sum, g2_nrm, a.norm_mode, a.eps);
            }
        } else if (cb == 0) {
            float *pp = a.partial + (size_t)(it.partoff + s) * a.part_stride + (size_t)a.C * TILE_PIX + tid;
            pp[0] = acc2[0];
            pp[TILE_PIX] = acc2[1];
            pp[2 * TILE_PIX] = acc2[2];
        }
        __syncthreads();                               // val4 is overwritten by the first value chunk
    }
    auto chunk = [&](float (&pre)[EPT_MAX][CHUNK], int c0) {
#if SLR_DBG & 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
        for (int j = 0; j < EPT_MAX; ++j)
#pragma unroll
            for (int h = 0; h < CHUNK / 4; ++h)
                val4[vslot<CHUNK>(tid + j * T, h)] = G2OK ? make_float4(pre[j][4 * h] * e_m[j], pre[j][4 * h + 1] * e_m[j], pre[j][4 * h + 2] * e_m[j], pre[j][4 * h + 3] * e_m[j])
                                                          : make_float4(pre[j][4 * h], pre[j][4 * h + 1], pre[j][4 * h + 2], pre[j][4 * h + 3]);
        if ((c0 - cb) / CHUNK < 9) SLR_STAMP(4 + 3 * ((c0 - cb) / CHUNK));
        __syncthreads();
#if SLR_DBG & 4
        bool dbg_bad = false;                          // staged value != what global memory holds?
        {
            float fresh[EPT_MAX][CHUNK];
            prefetch(fresh, c0);
#pragma unroll
            for (int j = 0; j < EPT_MAX; ++j)
#pragma unroll
                for (int u = 0; u < CHUNK; ++u) {
                    const float4 q = val4[vslot<CHUNK>(tid + j * T, u / 4)];
                    const float sv = (u & 3) == 0 ? q.x : (u & 3) == 1 ? q.y : (u & 3) == 2 ? q.z : q.w;
                    dbg_bad |= __float_as_uint(sv) != __float_as_uint(fresh[j][u]);
                }
        }
#endif
        if ((c0 - cb) / CHUNK < 9) SLR_STAMP(5 + 3 * ((c0 - cb) / CHUNK));
        prefetch(pre, c0 + 2 * CHUNK);                 // two chunks ahead (three: no gain fused, -20 % one flow: registers)
        float acc[CHUNK];
        gather(acc);
        // (a piece of a shared tile stores its raw sums into its partial slot: same addresses-by-selection, same NUMBER of store
        // instructions, only their scope differs -- an earlier form with one 16-byte inline-asm store was invisible to the
        // compiler's vmcnt bookkeeping, so every wait for a prefetched plane also waited for the write-through store's
        // acknowledgement, and a store path of its own made the counts differ where the paths merge)
        float rr[CHUNK];
        float *dd[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            float r = acc[u];
            float *dst = (c0 + u < cend) ? op + (size_t)(uint32_t)(c0 + u) * ostr : trash;
#ifdef SLR_CUT
            if (SLR_CUT == 5) dst = trash;   // (no output traffic)
#endif
            if (whole && !first) r = MAXOP ? fmaxf(r, *dst) : r + *dst;      // earlier segments of this tile
            if (NORM && single && last) r = finish(r, nrm, a.norm_mode, a.eps);
#if SLR_DBG & 4
            if (dbg_bad) r = 12345.0f;
#endif
            rr[u] = r; dd[u] = dst;
        }
        if (SCAN && part) {                            // agent-scope stores (sc1): read by another XCD inside this launch
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) store_wt4(dd[u], rr[u]);
        } else {
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) *dd[u] = rr[u];
        }
        if ((c0 - cb) / CHUNK < 9) SLR_STAMP(6 + 3 * ((c0 - cb) / CHUNK)); if (c0 + CHUNK >= cend) SLR_STAMP(40);
        __syncthreads();                               // val[] is overwritten by the next chunk
    };
    for (int c0 = cb; c0 < cend; c0 += 2 * CHUNK) {
        // queue state for the pop after this piece of work: ONE load, issued when the last two chunks begin, consumed after them
        if (SCAN && !ROWS && a.ctl && tid == 0 && last && !ctx && helper && c0 + 2 * CHUNK >= cend)
            qavail = (int)__hip_atomic_load((gu32 *)a.ctl + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chunk(preA, c0);
        if (c0 + CHUNK < cend) chunk(preB, c0 + CHUNK);
        SLR_CUT_AT(4);                                 // (after the first two chunks)
    }
  }
    if (ROWS) { SLR_STAMP(37); SLR_STAMP_RT(49); SLR_STAMP(38); SLR_STAMP_RT(50); }
    if (SCAN && !ROWS) {                     // scan front end: partial tiles of shared segments, then the next piece of work
        if (!w_piece) { SLR_STAMP(37); SLR_STAMP_RT(49); }
        if (part) {
            // every storing wave drains its write-through stores, then ONE arrival; the last segment to arrive combines
#if SLR_SHARE_STORE == 2
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                wsum[0] = atomicAdd(&a.arrive[(ROWS ? (size_t)w_grp * a.part_slots : (size_t)0) + w_po], 1u);   // (rows: one counter per channel group)
                // the partial slots were stored write-through and drained before their arrivals: ONE agent-scope acquire (drops this
                // CU's stale lines) and the combine below may use plain, pipelined loads

            }
            __syncthreads();
            const bool last_seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)wsum[0]) + 1u == ns_tile;
            __syncthreads();
            if (last_seg) {
                const int ly = tid / TILE_W, lx = tid - ly * TILE_W;
                const int oy = ty0 + ly, ox = tx0 + lx;
                const bool inside = (oy < a.H) & (ox < a.W);
                const float *pb = a.partial + (size_t)w_po * a.part_stride;
                const size_t pnorm = (size_t)((a.C + 3) / 4) * 4 * TILE_PIX;
                float nrm = 0.0f;
                float *o = a.out + (size_t)n * a.C * HW + (size_t)oy * a.W + ox;
                // agent-scope (sc1) loads, 8 in flight per wait (two pieces x four planes); pieces in fixed order: reproducible
                if (NORM) {
                    for (uint32_t q = 0; q < ns_tile; q += 4) {
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = load_sc1_4(pb + (size_t)min(q + i, ns_tile - 1u) * a.part_stride + pnorm + tid);
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
#pragma unroll
                        for (int i = 0; i < 4; ++i) nrm += q + i < ns_tile ? v[i] : 0.0f;
                    }
                    if (a.norm_out && inside && cb == 0) a.norm_out[(size_t)n * HW + (size_t)oy * a.W + ox] = norm_value(nrm, a.norm_mode, a.eps);
                }
                constexpr int CP = 4;                          // planes per round: 2 pieces x 4 planes = 8 loads in flight per wait (8: spills)
                for (int c0 = cb; c0 < cend; c0 += CP) {
                    float acc[CP];
                    for (uint32_t q = 0; q < ns_tile; q += 2) {
                        float v[2][CP];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int k = 0; k < CP; ++k)
                                v[i][k] = load_sc1_4(pb + (size_t)min(q + i, ns_tile - 1u) * a.part_stride + (size_t)min(c0 + k, cend - 1) * TILE_PIX + tid);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int k = 0; k < CP; ++k) {
                                asm volatile("" : "+v"(v[i][k]));          // (read only after the wait above)
                                if (q + i < ns_tile) acc[k] = q + i == 0 ? v[i][k] : MAXOP ? fmaxf(acc[k], v[i][k]) : acc[k] + v[i][k];
                            }
                    }
#pragma unroll
                    for (int k = 0; k < CP; ++k)
                        if (inside && c0 + k < cend) o[(size_t)(c0 + k) * HW] = NORM ? finish(acc[k], nrm, a.norm_mode, a.eps) : acc[k];
                }
            }
        }
        if (!a.ctl) break;
        // ---- next piece of work: claim a pushed segment (see the block comment above the kernel)
        if (tid == 0) {
            unsigned long long got = 0;
            const bool seek = ctx || (helper && qavail > 0);
            if (seek) {
#ifdef SLR_SCAN_STATS
                atomicAdd(&a.ctl[22], 1u);
#endif
                const int old = (int)atomicSub(&a.ctl[3], 1u);
                if (old <= 0) atomicAdd(&a.ctl[3], 1u);            // nothing on offer: give the claim back
                else {
                    const uint32_t idx = atomicAdd(&a.ctl[0], 1u);
                    for (uint32_t spin = 0; spin < (1u << 24); ++spin) {             // reserved by its writer, written next
                        got = __hip_atomic_load((gu64 *)(a.q_items + idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (got) break;
#ifdef SLR_SCAN_STATS
                        atomicAdd(&a.ctl[20], 1u);
#endif
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
            }
            wsum[0] = (uint32_t)got; wsum[1] = (uint32_t)(got >> 32);
        }
        __syncthreads();
        // (wave-uniform by construction: keep the claimed work in scalar registers, like the block's own tile)
        const unsigned long long wv = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)wsum[1]) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)wsum[0]);
        __syncthreads();
        SLR_STAMP(38); SLR_STAMP_RT(50);
        if (wv == 0) break;
#ifdef SLR_TRACE
        if (a.trace && tid == 0) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLR_TRACE_SLOTS + 39] += 1;
#endif
        w_piece = true;
        {
            item = (uint32_t)(wv >> 38) - 1u;
            w_grp = (uint32_t)wv & 0xfu;
            w_seg = (uint32_t)(wv >> 30) & 0xffu;
            w_po = (uint32_t)(wv >> 4) & 0x3ffffu;
        }
        continue;
    }
    if (!WHOLE) break;
    wi += gridDim.x;
    if (wi >= a.totals[4]) break;
    item = a.whole_items[wi];
  }
}

// Multi-segment tiles: sum (max) the raw partial tiles in segment order, normalise, store.
// grid (max multi-segment tiles, ceil(C/COMBINE_CHUNK)); TILE_PIX threads; workgroups past the
// number of multi-segment tiles of this plan (totals[3]) exit at once.
template <bool NORM, bool MAXOP>
__global__ __launch_bounds__(SPLAT_THREADS) void combine_kernel(SplatBatch batch) {
    uint32_t bstart;
    const SplatArgs &a = batch.f[batch_frame(batch.cend, batch.nb, blockIdx.x, bstart)];
    const uint32_t bx = blockIdx.x - bstart;
    if (bx >= a.totals[3]) return;
    const uint32_t t = a.multi[bx];
    const uint32_t ns = a.nseg[t];
    const int c0 = blockIdx.y * COMBINE_CHUNK;
    const int n = t / a.tiles, tl = t - n * a.tiles;
    const int ty0 = (tl / a.tiles_x) * TILE_H, tx0 = (tl % a.tiles_x) * TILE_W;
    const int HW = a.H * a.W;
    const int tid = threadIdx.x;
    const int ly = tid / TILE_W, lx = tid - ly * TILE_W;
    const int oy = ty0 + ly, ox = tx0 + lx;
    if (oy >= a.H || ox >= a.W) return;
    const float *src = a.partial + (size_t)a.partoff[t] * a.part_stride + tid;
    float nrm = 0.0f;
    if (NORM) {
        for (uint32_t s = 0; s < ns; ++s) nrm += src[(size_t)s * a.part_stride + (size_t)a.C * TILE_PIX];
        if (a.norm_out && blockIdx.y == 0)
            a.norm_out[(size_t)n * HW + (size_t)oy * a.W + ox] = norm_value(nrm, a.norm_mode, a.eps);
        if (a.in2 && blockIdx.y == 0) {                            // second weight group: raw sum and normaliser planes C+1, C+2
            float s2 = 0.0f, n2 = 0.0f;
            for (uint32_t s = 0; s < ns; ++s) {
                s2 += src[(size_t)s * a.part_stride + (size_t)(a.C + 1) * TILE_PIX];
                n2 += src[(size_t)s * a.part_stride + (size_t)(a.C + 2) * TILE_PIX];
            }
            a.out2[(size_t)n * HW + (size_t)oy * a.W + ox] = finish(s2, n2, a.norm_mode, a.eps);
        }
    }
    float *op = a.out + (size_t)n * a.C * HW + (size_t)oy * a.W + ox;
    for (int c = c0; c < min(c0 + COMBINE_CHUNK, a.C); ++c) {
        float acc = src[(size_t)c * TILE_PIX];
        for (uint32_t s = 1; s < ns; ++s) {                       // fixed order: deterministic
            const float v = src[(size_t)s * a.part_stride + (size_t)c * TILE_PIX];
            acc = MAXOP ? fmaxf(acc, v) : acc + v;
        }
        if (NORM) acc = finish(acc, nrm, a.norm_mode, a.eps);
        op[(size_t)c * HW] = acc;
    }
}

// =========================================================================== small kernels

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

struct Ws {
    WsLayout L;
    char *base;
    uint32_t *count, *cursor, *listoff, *list, *nseg, *partoff, *totals, *multi, *whole_items;
    ItemDesc *items;
    float *partial, *trash;
    SrcBox *box;
    uint32_t *ctl, *arrive;
    unsigned long long *queue, *rowcnt;
    RowRec *rowlist;
};

static int ws_open(Ws &w, int N, int C, int H, int W, void *ws, size_t bytes, const char *who) {
    w.L = ws_layout(N, C, H, W);
    if (!ws || ((uintptr_t)ws & 15) || bytes < w.L.total) {
        set_error("%s: workspace needs %zu bytes (16-byte aligned), got %zu", who, w.L.total, bytes);
        return SLR_E_WORKSPACE;
    }
    w.base = (char *)ws;
    w.count = (uint32_t *)(w.base + w.L.off_count);
    w.cursor = (uint32_t *)(w.base + w.L.off_cursor);
    w.listoff = (uint32_t *)(w.base + w.L.off_listoff);
    w.list = (uint32_t *)(w.base + w.L.off_list);
    w.nseg = (uint32_t *)(w.base + w.L.off_nseg);
    w.partoff = (uint32_t *)(w.base + w.L.off_partoff);
    w.multi = (uint32_t *)(w.base + w.L.off_multi);
    w.whole_items = (uint32_t *)(w.base + w.L.off_whole);
    w.items = (ItemDesc *)(w.base + w.L.off_items);
    w.totals = (uint32_t *)(w.base + w.L.off_totals);
    w.partial = (float *)(w.base + w.L.off_partial);
    w.trash = (float *)(w.base + w.L.off_trash);
    w.box = (SrcBox *)(w.base + w.L.off_box);
    w.ctl = (uint32_t *)(w.base + w.L.off_ctl);
    w.queue = (unsigned long long *)(w.base + w.L.off_queue);
    w.arrive = (uint32_t *)(w.base + w.L.off_arrive);
    w.rowcnt = (unsigned long long *)(w.base + w.L.off_rowcnt);
    w.rowlist = (RowRec *)(w.base + w.L.off_rowlist);
    return 0;
}

static int check_dims(int N, int C, int H, int W, const char *who) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)N * H * W >= (1LL << 29)) {
        set_error("%s: bad sizes N=%d C=%d H=%d W=%d", who, N, C, H, W);
        return SLR_E_BADARG;
    }
    return 0;
}

// Bin one flow (w1 == nullptr) or two flows of the same shape into their workspaces.
static int do_bin(const float *flow0, Ws &w0, const float *flow1, Ws *w1, int N, int H, int W, hipStream_t st) {
    const int nf = w1 ? 2 : 1;
    BinSet b = {};
    Ws *ws[2] = {&w0, w1};
    const float *fl[2] = {flow0, flow1};
    for (int k = 0; k < nf; ++k) {
        b.flow[k] = fl[k];
        b.count[k] = ws[k]->count; b.cursor[k] = ws[k]->cursor; b.listoff[k] = ws[k]->listoff; b.list[k] = ws[k]->list;
    }
    // the per-tile counts of both flows are zeroed by one launch of our own (not hipMemsetAsync: one launch
    // instead of two runtime fill kernels, and a captured hipMemsetAsync node on a workspace from torch's graph-private
    // pool made HIP-graph replays of the binning fault -- tools/dev/graph_try.py, test_frame_is_graph_capturable)
    hipLaunchKernelGGL(zero_counts_kernel, dim3((w0.L.nt + 255) / 256, nf), dim3(256), 0, st, b, w0.L.nt);
    dim3 grid((H * W + 256 * BIN_PPT - 1) / (256 * BIN_PPT), N, nf);
    hipLaunchKernelGGL(bin_kernel<false>, grid, dim3(256), 0, st, b, H, W, w0.L.tiles_x, w0.L.tiles);
    hipLaunchKernelGGL(offsets_kernel, dim3(nf), dim3(1024), 0, st, b, w0.L.nt);
    hipLaunchKernelGGL(bin_kernel<true>, grid, dim3(256), 0, st, b, H, W, w0.L.tiles_x, w0.L.tiles);
    SLR_CHECK_LAUNCH();
    return 0;
}

template <bool NORM, bool MAXOP, int EPT, int CHUNK, bool WHOLE, int FE = 0>
static int launch_tile_variant(const SplatBatch &b, uint32_t grid, uint32_t groups, size_t lds, hipStream_t st) {
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device (a process may drive
    // several GPUs, e.g. the DataParallel replicas of the reference's training scripts)
    static bool attr_set[64] = {};
    int dev = 0;
    SLR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        SLR_CHECK_HIP(hipFuncSetAttribute((const void *)splat_tile_kernel<NORM, MAXOP, EPT, CHUNK, WHOLE, FE>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((splat_tile_kernel<NORM, MAXOP, EPT, CHUNK, WHOLE, FE>), dim3(grid, groups), dim3(SPLAT_THREADS), lds, st, b);
    return 0;
}

// What the host knows about the plan of one frame (a clip plan's totals read back once per clip); < 0 = unknown ->
// the grids cover the upper bounds and surplus workgroups exit at once.
struct PlanHint { int n_items, n_multi, n_whole; };

static uint32_t channel_groups(uint32_t nt, int C, int chunk) {
    if (SLR_CSPLIT_MAX <= 1) return 1u;
    const uint32_t fit = SLR_CSPLIT_SLOTS / (nt ? nt : 1u), byc = (uint32_t)C / (2u * chunk);
    uint32_t groups = fit < (uint32_t)SLR_CSPLIT_MAX ? fit : (uint32_t)SLR_CSPLIT_MAX;
    groups = groups < byc ? groups : byc;
    return groups < 1u ? 1u : groups;
}

// Tile kernel, whole-tile kernel and combine for nb frames whose plans are already in b.f[]: the main tile kernel
// and combine as ONE launch each over all frames.
template <bool NORM, bool MAXOP, int EPT, int CHUNK>
static int launch_batch(SplatBatch &b, const PlanHint *hint, uint32_t items_cap, uint32_t nt, uint32_t part_slots,
                        hipStream_t st) {
    // counts (T words) + wave sums (16) + offsets (T halfwords) | records | CHUNK staged planes + the all-zero NULL entry
    const size_t lds = lds_head_bytes(EPT) + (size_t)CHUNK * (EPT * SPLAT_THREADS + 1) * 4 + SLR_LDS_PAD;
    // Frames of a batch are consecutive frames of a clip: tile T of frame k+1 gathers from almost the same source region
    // as tile T of frame k (the displacement maps differ by one Euler step).  With the block ranges of the frames laid
    // end to end, those two workgroups run a whole frame apart and both fetch the region from HBM; with the groups of
    // 8 * XCD_GROUP blocks dealt round-robin over the frames they run side by side on the same XCD and share its L2.
    const bool interleave = SLR_BATCH_INTERLEAVE && b.nb > 1;
    b.interleave = interleave ? 1u : 0u;
    uint32_t grid = 0, cgrid = 0, gmax = 0;
    for (uint32_t i = 0; i < b.nb; ++i) {
        const int ni = hint[i].n_items, nm = hint[i].n_multi;
        const uint32_t cover = ni >= 0 && (uint32_t)ni < items_cap ? (uint32_t)ni : items_cap;
        const uint32_t own = ((cover + 8 * XCD_GROUP - 1) / (8 * XCD_GROUP)) * 8 * XCD_GROUP;
        grid += own;
        gmax = own > gmax ? own : gmax;
        b.end[i] = interleave ? own : grid;
        // every multi-segment tile owns >= 2 partial slots -> at most part_slots / 2 of them
        cgrid += nm >= 0 ? (uint32_t)nm : part_slots / 2;
        b.cend[i] = cgrid;
    }
    if (interleave) grid = gmax * b.nb;              // (frames with fewer groups leave a few empty blocks)
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));       // slr_splat_time_next:
    // Small grids (fewer tiles than the chip has workgroup slots: 256 CUs x 2): a tile is one workgroup whose chunk
    // pipeline nothing overlaps with, so the planes are dealt to 2-4 workgroups per tile (gridDim.y; each builds the
    // tile's records itself, phase 1 is the cheap part).  256x480, C = 64 (config C2): 32 -> ... us.
    const uint32_t groups = b.nb == 1 ? channel_groups(nt, b.f[0].C, CHUNK) : 1u;
    if (grid)
        if (int e = launch_tile_variant<NORM, MAXOP, EPT, CHUNK, false>(b, grid, groups, lds, st)) return e;
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));         // the dominant kernel only
    g_ev_start = g_ev_stop = nullptr;                                                // one-shot
    // tiles that did not fit the partial-slot budget (none for ordinary flows), frame by frame
    for (uint32_t i = 0; i < b.nb; ++i) {
        if (hint[i].n_whole == 0) continue;
        SplatBatch one = {};
        one.f[0] = b.f[i];
        one.nb = 1;
        const uint32_t wg = hint[i].n_whole > 0 ? (uint32_t)hint[i].n_whole : nt;
        one.end[0] = wg < 256u ? wg : 256u;
        if (int e = launch_tile_variant<NORM, MAXOP, EPT, CHUNK, true>(one, one.end[0], 1u, lds, st)) return e;
    }
    if (cgrid)
        hipLaunchKernelGGL((combine_kernel<NORM, MAXOP>), dim3(cgrid, (b.f[0].C + COMBINE_CHUNK - 1) / COMBINE_CHUNK),
                           dim3(SPLAT_THREADS), 0, st, b);
    SLR_CHECK_LAUNCH();
    return 0;
}

template <bool NORM, bool MAXOP>
static int run_batch(SplatBatch &b, const PlanHint *hint, bool two_flows, uint32_t items_cap, uint32_t nt,
                     uint32_t part_slots, hipStream_t st) {
    for (uint32_t i = 0; i < b.nb; ++i) {
        b.f[i].ndir = two_flows ? 2 : 1;
        b.f[i].seg = two_flows ? SEG_TWO : SEG_ONE;
#ifdef SLR_TRACE
        b.f[i].trace = g_trace;
#endif
    }
    if (two_flows) return launch_batch<NORM, MAXOP, EPT_TWO, CHUNK_TWO>(b, hint, items_cap, nt, part_slots, st);
    return launch_batch<NORM, MAXOP, EPT_ONE, CHUNK_ONE>(b, hint, items_cap, nt, part_slots, st);
}

// tile kernel(s) + combine for ONE frame whose plan is already in `a`.
template <bool NORM, bool MAXOP>
static int run_plan(SplatArgs &a, bool two_flows, uint32_t items_cap, uint32_t nt, uint32_t part_slots, int n_items,
                    int n_multi, int n_whole, hipStream_t st) {
    SplatBatch b = {};
    b.f[0] = a;
    b.nb = 1;
    const PlanHint h = {n_items, n_multi, n_whole};
    return run_batch<NORM, MAXOP>(b, &h, two_flows, items_cap, nt, part_slots, st);
}

#ifdef SLR_SCAN_ORDER_HOOK
static const uint32_t *g_scan_order = nullptr;
#endif
static std::atomic<int> g_scan_max_tiles{SLR_SCAN_MAX_TILES};          // slr_splat_set_scan_max_tiles
static std::atomic<int> g_front_end{SLR_FRONT_END};                    // slr_splat_set_front_end
// Front end of a one-flow call that brings no bins: 0 bins, 1 scan (boxes), 2 rows.  By default the grid decides: scan up to
// slr_splat_set_scan_max_tiles tiles (single-round grids: no plan to wait for), rows above.
static int front_end(int prebinned, uint32_t nt, int H) {
    if (prebinned) return 0;
    int fe = g_front_end.load(std::memory_order_relaxed);
    if (fe < 0 || fe > 2) fe = nt <= (uint32_t)g_scan_max_tiles.load(std::memory_order_relaxed) ? 1 : 2;
    if (fe == 2 && H >= (1 << 24)) fe = 0;             // (a row-list entry keeps the image row in 24 bits)
    return fe;
}

// The scan front end of a one-flow call: box kernel + tile kernel, nothing else (no bins, no plan, no combine).
template <bool NORM, bool MAXOP>
static int do_splat_scan(SplatArgs a, Ws &w0, hipStream_t st) {
    a.tiles_x = w0.L.tiles_x;
    a.tiles = w0.L.tiles;
    a.trash = w0.trash;
    a.box = w0.box;
    a.nt = w0.L.nt;
    a.partial = w0.partial;
    a.part_stride = w0.L.part_stride;
    a.part_slots = w0.L.part_slots;
    a.ctl = SLR_SCAN_SHARE ? w0.ctl : nullptr;
    a.q_items = w0.queue;
    a.arrive = w0.arrive;
    a.ndir = 1;
    a.seg = EPT_SCAN * SPLAT_THREADS;
#ifdef SLR_TRACE
    a.trace = g_trace;
#endif
#ifdef SLR_SCAN_ORDER_HOOK
    a.order = g_scan_order;
#endif
    hipLaunchKernelGGL(scan_box_kernel, dim3(w0.L.nt), dim3(TILE_PIX), 0, st, a.flow[0], w0.box, a.H, a.W, w0.L.tiles_x, w0.L.tiles,
                       w0.ctl, w0.queue, w0.arrive, w0.L.part_slots);
    SplatBatch b = {};
    b.f[0] = a;
    b.nb = 1;
    const uint32_t grid = ((w0.L.nt + 8 * XCD_GROUP - 1) / (8 * XCD_GROUP)) * 8 * XCD_GROUP;
    b.end[0] = grid;
    const size_t lds = lds_head_bytes(EPT_SCAN, true) + (size_t)CHUNK_ONE * (EPT_SCAN * SPLAT_THREADS + 1) * 4 + SLR_LDS_PAD;
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));
    if (int e = launch_tile_variant<NORM, MAXOP, EPT_SCAN, CHUNK_ONE, false, 1>(b, grid, channel_groups(w0.L.nt, a.C, CHUNK_ONE), lds, st)) return e;
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
    g_ev_start = g_ev_stop = nullptr;
    SLR_CHECK_LAUNCH();
    return 0;
}

// The rows front end of a one-flow call: zero + rowbin (with the plan) + tile kernel (multi-segment tiles summed in-kernel).
template <bool NORM, bool MAXOP>
static int do_splat_rows(SplatArgs a, Ws &w0, hipStream_t st) {
    a.tiles_x = w0.L.tiles_x;
    a.tiles = w0.L.tiles;
    a.trash = w0.trash;
    a.nt = w0.L.nt;
    a.partial = w0.partial;
    a.part_stride = w0.L.part_stride;
    a.part_slots = w0.L.part_slots;
    a.ctl = nullptr;
    a.arrive = w0.arrive;
    a.rowlist = w0.rowlist;
    a.items_cap = w0.L.rows_items_cap;
    a.items = w0.items;
    a.totals = w0.totals;
    a.whole_items = w0.whole_items;
    a.ndir = 1;
    a.seg = EPT_SCAN * SPLAT_THREADS;
#ifdef SLR_TRACE
    a.trace = g_trace;
#endif
    const uint32_t nt = w0.L.nt;
    hipLaunchKernelGGL(rows_zero_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, w0.rowcnt, nt, w0.ctl, (uint32_t *)w0.box);   // (the box array of
    // the scan front end doubles as rowbin_kernel's first-level arrival counters)
    const uint32_t rb_grid = (uint32_t)a.N * (uint32_t)w0.L.tiles_x * (uint32_t)((w0.L.tiles_y + ROWBIN_R - 1) / ROWBIN_R);
    hipLaunchKernelGGL(rowbin_kernel, dim3(rb_grid), dim3(TILE_PIX), 0, st, a.flow[0], w0.rowcnt, w0.rowlist, a.H, a.W, w0.L.tiles_x,
                       w0.L.tiles_y, nt, w0.ctl, (uint32_t *)w0.box, (uint32_t)a.seg, nt > 512u ? (uint32_t)SLR_PLAN_HEAVY : 0u,
                       w0.L.rows_items_cap, w0.items, w0.totals);
    SplatBatch b = {};
    b.f[0] = a;
    b.nb = 1;
    // (the grid covers the bound on the plan's items -- 4 per tile; workgroups past totals[0] exit at once: measured free,
    // 1920 vs 3840 blocks on the identity flow 142.9 vs 142.3 us)
    const uint32_t grid = ((w0.L.rows_items_cap + 8 * XCD_GROUP - 1) / (8 * XCD_GROUP)) * 8 * XCD_GROUP;
    b.end[0] = grid;
    const size_t lds = lds_head_bytes(EPT_SCAN, true) + (size_t)CHUNK_ONE * (EPT_SCAN * SPLAT_THREADS + 1) * 4 + SLR_LDS_PAD;
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));
    if (int e = launch_tile_variant<NORM, MAXOP, EPT_SCAN, CHUNK_ONE, false, 2>(b, grid, channel_groups(nt, a.C, CHUNK_ONE), lds, st)) return e;
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
    g_ev_start = g_ev_stop = nullptr;
    // pieces that hold more than SEG entries (none for ordinary flows; appended by their workgroups above): pass by pass
    // (their planes dealt to up to 8 workgroups each: these run after everybody else, on an empty chip, one pass after the other)
    b.end[0] = nt < 64u ? nt : 64u;                     // (grid-strided over the list; normally it is empty)
    const uint32_t wgroups = (uint32_t)a.C / (2u * CHUNK_ONE) < 1u ? 1u : (uint32_t)a.C / (2u * CHUNK_ONE) > 8u ? 8u : (uint32_t)a.C / (2u * CHUNK_ONE);
    // (and with passes of 2048 entries -- 86 KiB of LDS, one workgroup per CU: most of these pieces are just over the 1024 of the
    // main kernel and finish in one pass)
    constexpr int EPT_DEFER = SLR_EPT_DEFER;
    b.f[0].seg = EPT_DEFER * SPLAT_THREADS;
    const size_t lds_defer = lds_head_bytes(EPT_DEFER, true) + (size_t)CHUNK_ONE * (EPT_DEFER * SPLAT_THREADS + 1) * 4 + SLR_LDS_PAD;
    if (int e = launch_tile_variant<NORM, MAXOP, EPT_DEFER, CHUNK_ONE, true, 2>(b, b.end[0], wgroups, lds_defer, st)) return e;
    SLR_CHECK_LAUNCH();
    return 0;
}

// plan + splat + combine.  w0 holds the plan and the partial tiles; w1 (optional) the second bin.
template <bool NORM, bool MAXOP>
static int do_splat(SplatArgs a, Ws &w0, Ws *w1, hipStream_t st) {
    a.tiles_x = w0.L.tiles_x;
    a.tiles = w0.L.tiles;
    a.count[0] = w0.count; a.listoff[0] = w0.listoff; a.list[0] = w0.list;
    a.count[1] = w1 ? w1->count : nullptr; a.listoff[1] = w1 ? w1->listoff : nullptr; a.list[1] = w1 ? w1->list : nullptr;
    a.nseg = w0.nseg; a.partoff = w0.partoff; a.items = w0.items; a.totals = w0.totals; a.multi = w0.multi; a.whole_items = w0.whole_items;
    a.partial = w0.partial;
    a.trash = w0.trash;
    a.part_stride = w0.L.part_stride;
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)w0.count,
                       (const uint32_t *)(w1 ? w1->count : nullptr), (const uint32_t *)w0.listoff,
                       (const uint32_t *)(w1 ? w1->listoff : nullptr), w0.L.nt, (uint32_t)w0.L.tiles_x,
                       (uint32_t)w0.L.tiles_y, (uint32_t)(w1 ? SEG_TWO : SEG_ONE),
                       w0.L.part_slots, w0.nseg, w0.partoff, w0.items, w0.multi, w0.whole_items, w0.totals);
    return run_plan<NORM, MAXOP>(a, w1 != nullptr, w0.L.items_cap, w0.L.nt, w0.L.part_slots, -1, -1, -1, st);
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

#ifdef SLR_SCAN_STATS
SLR_EXPORT size_t slr_debug_ctl_offset(int N, int C, int H, int W) { return ws_layout(N, C, H, W).off_ctl; }
#endif

#ifdef SLR_PLAN_STAMPS
SLR_EXPORT size_t slr_debug_totals_offset(int N, int C, int H, int W) { return ws_layout(N, C, H, W).off_totals; }
SLR_EXPORT size_t slr_debug_rowcnt_offset(int N, int C, int H, int W) { return ws_layout(N, C, H, W).off_rowcnt; }
#endif

#ifdef SLR_SCAN_ORDER_HOOK
SLR_EXPORT void slr_debug_scan_order(const uint32_t *order) { slr::g_scan_order = order; }
#endif

SLR_EXPORT int slr_splat_set_scan_max_tiles(int max_tiles) {
    return slr::g_scan_max_tiles.exchange(max_tiles < 0 ? 0 : max_tiles);
}

SLR_EXPORT int slr_splat_set_front_end(int front_end) {
    return slr::g_front_end.exchange(front_end < 0 || front_end > 2 ? -1 : front_end);
}

SLR_EXPORT size_t slr_splat_workspace_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C < 0 || H <= 0 || W <= 0) return 0;
    return ws_layout(N, C, H, W).total;
}

SLR_EXPORT int slr_splat_bin(const float *flow, int N, int C, int H, int W, void *ws, size_t ws_bytes,
                             void *stream) {
    SLR_CHECK_ARG(flow, "null flow");
    if (int e = check_dims(N, C > 0 ? C : 1, H, W, __func__)) return e;
    Ws w;
    if (int e = ws_open(w, N, C, H, W, ws, ws_bytes, __func__)) return e;
    return do_bin(flow, w, nullptr, nullptr, N, H, W, (hipStream_t)stream);
}

SLR_EXPORT int slr_splat_bin_pair(const float *flow_a, const float *flow_b, int N, int C, int H, int W, void *ws_a,
                                  void *ws_b, size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(flow_a && flow_b, "null flow");
    SLR_CHECK_ARG(ws_a != ws_b, "the two flows need separate workspaces");
    if (int e = check_dims(N, C > 0 ? C : 1, H, W, __func__)) return e;
    Ws wa, wb;
    if (int e = ws_open(wa, N, C, H, W, ws_a, ws_bytes, __func__)) return e;
    if (int e = ws_open(wb, N, C, H, W, ws_b, ws_bytes, __func__)) return e;
    return do_bin(flow_a, wa, flow_b, &wb, N, H, W, (hipStream_t)stream);
}

SLR_EXPORT int slr_softsplat_forward(const float *in, const float *flow, float *out, int N, int C, int H,
                                     int W, void *ws, size_t ws_bytes, int prebinned, void *stream) {
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    if (int e = check_dims(N, C, H, W, __func__)) return e;
    Ws w;
    if (int e = ws_open(w, N, C, H, W, ws, ws_bytes, __func__)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int fe = front_end(prebinned, w.L.nt, H);
    if (!prebinned && fe == 0) if (int e = do_bin(flow, w, nullptr, nullptr, N, H, W, st)) return e;
    SplatArgs a = {};
    a.in = in; a.flow[0] = flow; a.scale[0] = 1.0f; a.out = out;
    a.N = N; a.C = C; a.H = H; a.W = W; a.mulmode = MUL_ONE;
    if (fe == 1) return do_splat_scan<false, false>(a, w, st);
    if (fe == 2) return do_splat_rows<false, false>(a, w, st);
    return do_splat<false, false>(a, w, nullptr, st);
}

SLR_EXPORT int slr_softsplat_mode_forward(const float *in, const float *metric, const float *flow, float *out,
                                          int N, int C, int H, int W, int mode, void *ws, size_t ws_bytes,
                                          int prebinned, void *stream) {
    SLR_CHECK_ARG(mode >= SLR_MODE_SUMMATION && mode <= SLR_MODE_SOFTMAX, "mode");
    if (mode == SLR_MODE_SUMMATION)
        return slr_softsplat_forward(in, flow, out, N, C, H, W, ws, ws_bytes, prebinned, stream);
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    SLR_CHECK_ARG(mode == SLR_MODE_AVERAGE || metric, "metric required for linear/softmax");
    if (int e = check_dims(N, C, H, W, __func__)) return e;
    Ws w;
    if (int e = ws_open(w, N, C, H, W, ws, ws_bytes, __func__)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int fe = front_end(prebinned, w.L.nt, H);
    if (!prebinned && fe == 0) if (int e = do_bin(flow, w, nullptr, nullptr, N, H, W, st)) return e;
    SplatArgs a = {};
    a.in = in; a.mul = metric; a.flow[0] = flow; a.scale[0] = 1.0f; a.out = out;
    a.N = N; a.C = C; a.H = H; a.W = W;
    a.mulmode = mode == SLR_MODE_AVERAGE ? MUL_ONE : mode == SLR_MODE_LINEAR ? MUL_PLANE : MUL_EXP;
    a.norm_mode = SLR_NORM_ZERO_TO_ONE;
    if (fe == 1) return do_splat_scan<true, false>(a, w, st);
    if (fe == 2) return do_splat_rows<true, false>(a, w, st);
    return do_splat<true, false>(a, w, nullptr, st);
}

SLR_EXPORT int slr_synth_group(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                               const float *disp_f, const float *disp_p, float alpha, float *out,
                               float *norm_out, int C, int H, int W, float eps, void *ws_f, void *ws_p,
                               size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(values && wlogit && disp_f && disp_p && out, "null pointer");
    if (int e = check_dims(1, C, H, W, __func__)) return e;
    Ws wf, wp;
    if (int e = ws_open(wf, 1, C, H, W, ws_f, ws_bytes, __func__)) return e;
    if (int e = ws_open(wp, 1, C, H, W, ws_p, ws_bytes, __func__)) return e;
    SplatArgs a = {};
    a.in = values; a.mul = wlogit; a.mulmax = wmax;
    a.flow[0] = disp_f; a.flow[1] = disp_p;
    a.scale[0] = alpha; a.scale[1] = 1.0f - alpha;
    a.out = out; a.norm_out = norm_out;
    a.N = 1; a.C = C; a.H = H; a.W = W;
    a.mulmode = wmax ? MUL_EXP_SHIFT : (exp_weights ? MUL_EXP : MUL_PLANE);
    a.norm_mode = SLR_NORM_CLAMP_EPS;
    a.eps = eps;
    return do_splat<true, false>(a, wf, &wp, (hipStream_t)stream);
}

SLR_EXPORT int slr_splat_normalize(const float *accum, float *out, int N, int C, int H, int W, int norm_mode,
                                   float eps, void *stream) {
    SLR_CHECK_ARG(accum && out, "null pointer");
    SLR_CHECK_ARG(norm_mode == SLR_NORM_ZERO_TO_ONE || norm_mode == SLR_NORM_CLAMP_EPS, "norm_mode");
    if (int e = check_dims(N, C, H, W, __func__)) return e;
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(normalize_kernel, grid, dim3(256), 0, (hipStream_t)stream, accum, out, C, H * W,
                       norm_mode, eps);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_global_max(const float *x, size_t n, float *result, float *scratch, void *stream) {
    SLR_CHECK_ARG(x && result && scratch && n > 0, "null pointer / empty");
    int blocks = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(max_stage_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
    hipLaunchKernelGGL(max_stage_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float *)scratch,
                       (size_t)blocks, result);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_maxsplat_forward(const float *in, const float *flow, float *out, float init, int N, int C,
                                    int H, int W, void *ws, size_t ws_bytes, int prebinned, void *stream) {
    SLR_CHECK_ARG(in && flow && out, "null pointer");
    if (int e = check_dims(N, C, H, W, __func__)) return e;
    Ws w;
    if (int e = ws_open(w, N, C, H, W, ws, ws_bytes, __func__)) return e;
    hipStream_t st = (hipStream_t)stream;
    const int fe = front_end(prebinned, w.L.nt, H);
    if (!prebinned && fe == 0) if (int e = do_bin(flow, w, nullptr, nullptr, N, H, W, st)) return e;
    SplatArgs a = {};
    a.in = in; a.flow[0] = flow; a.scale[0] = 1.0f; a.out = out; a.init = init;
    a.N = N; a.C = C; a.H = H; a.W = W; a.mulmode = MUL_ONE;
    if (fe == 1) return do_splat_scan<false, true>(a, w, st);
    if (fe == 2) return do_splat_rows<false, true>(a, w, st);
    return do_splat<false, true>(a, w, nullptr, st);
}

