// splat_clip.hip -- the splat stage of a CLIP: all frames' displacement maps binned by ROW SEGMENTS once per clip, then ONE fused
// kernel per batch of frames does what forward_flow does between encoder and decoder for every frame of the batch
// (models/animating_softmax_splating.py:847-924, ..._2layers_alpha_seperate.py:950-1045): exp-weighting, both splat directions,
// the 2-layer model's alpha plane as a second weight group, normalisation.
//
//   per clip   rowbin_clip_kernel   every 64-pixel row segment of every displacement map is appended to the few 8x64 OUTPUT tiles its
//                                   footprints touch: one returning 64-bit atomic per (segment, tile) = list slot + exact entry count,
//                                   two more add the segment's hits per column octant (exact, 16 bits each);
//              rows_plan_clip_kernel one workgroup per frame turns the counts of its two maps into work items: a tile of at most
//                                   SEG entries is one item, a heavier one is cut into ranges of its OUTPUT COLUMNS (whole octants,
//                                   by the exact histogram) -- a piece stages every entry that touches its columns and owns its
//                                   output pixels: no partial tiles, no combine pass, nothing summed across workgroups;
//   per batch  clip_tile_kernel     one workgroup = one piece of one frame (up to 16 frames per launch, their block groups
//                                   interleaved so the same tile of consecutive frames shares an XCD's L2): the two row-segment
//                                   lists -> the flow rows (coalesced) -> entries in LDS -> per-output-pixel records -> chunk pipeline;
//              clip_tile_kernel<.., PASSES>  a piece that still holds more than SEG entries (an octant that is a sink by itself; any
//                                   pathological flow) was appended to the frame's deferred list by its workgroup; this normally
//                                   empty launch walks such pieces pass by pass.
// Round 3 did this with per-pixel bins (two passes over all maps with one atomic per footprint: 29 us per frame amortised), partial
// tiles for multi-segment tiles and a combine pass (17 us per frame): stage 230 us per frame around a 180 us kernel.
#include "splat_rows.hpp"
#include "splat_ws.hpp"


namespace slr {

using ClipCfg = TileCfg<2, SLR_EPT_TWO, false, SLR_KREG_TWO>;      // two flows: 1536 entries per workgroup, 8-byte records, 79 KiB of LDS
using ClipPassCfg = TileCfg<2, SLR_EPT_DEFER, false, SLR_KREG_TWO>; // the pass-by-pass launch: passes of 2048 entries, 103 KiB
using ClipCfgB4 = TileCfg<2, SLR_EPT_TWO, false, SLR_KREG_TWO, true>;         // ... with the value planes plane-blocked by 4 (slr_pack_planes4)
using ClipPassCfgB4 = TileCfg<2, SLR_EPT_DEFER, false, SLR_KREG_TWO, true>;
constexpr int CT = TT;                             // work-items per workgroup = output pixels of a tile
constexpr int C_SEG = ClipCfg::SEG;                // entries a workgroup stages at once
constexpr int C_MAXB = SLR_CLIP_MAXB;              // frames per launch
constexpr int C_XCD = SLR_XCD_GROUP;
constexpr uint32_t CLIP_TOTALS = 8;                // per frame: [0] items, [4] deferred pieces, [5] arrivals of the deferred launch
constexpr uint32_t C_DEFER_WG = 16;                // workgroups per frame of the deferred launch
static_assert(2 * ClipCfg::NDIR * ROW_CAP * 4 + C_SEG * 8 <= ClipCfg::REC_BYTES, "the row lists and the second group's entry words live in the record area");

struct ClipBatch {
    TileShared s;
    TileFrame f[C_MAXB];
    uint32_t nb, interleave;
};
static_assert(sizeof(ClipBatch) <= 4096, "kernel arguments are limited to 4 KiB");

// =========================================================================== per clip: row segments of every map -> tiles
struct ClipRows {
    const float *disp[2];          // [*,2,H,W] displacement maps of the two directions
    const int *idx[2];             // [nframes] which map of disp[d] frame i uses
    unsigned long long *rowcnt;    // [2 * nframes][nt][4]
    RowRec *rowlist;               // [2 * nframes][nt][ROW_CAP]
    uint32_t nframes, nt;
};

__global__ __launch_bounds__(256) void zero_u64_kernel(unsigned long long *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// grid (tiles_x * ceil(tiles_y / 2), 2 * nframes): a workgroup covers two vertically adjacent source tiles of one map (wave w: rows w
// and w + 8 of the block, both flow loads in flight).  The distinct tiles a row's 64 footprints touch are found by ballots (the
// first lane with something left names a tile, a ballot counts the lanes that touch it); round k's append is parked in lane k and all
// appends of the wave go out as one set of atomic instructions.
__global__ __launch_bounds__(CT) void rowbin_clip_kernel(ClipRows r, int H, int W, int tiles_x, int tiles_y) {
    constexpr int R = SLR_ROWBIN_CLIP_R;
    const uint32_t m = blockIdx.y, d = m >= r.nframes ? 1u : 0u, fi = m - d * r.nframes;
    const float *fl = r.disp[d] + (size_t)r.idx[d][fi] * 2 * H * W;
    const int bl = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stx = bl % tiles_x, y_base = (bl / tiles_x) * R * TILE_H + wid, x = stx * TILE_W + lane;
    float fx[R], fy[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int y = y_base + k * TILE_H;
        const size_t q = (y < H && x < W) ? (size_t)y * W + x : 0;
        fx[k] = fl[q];
        fy[k] = fl[(size_t)H * W + q];
    }
    unsigned long long *cnt_m = r.rowcnt + (size_t)m * r.nt * 4;
    RowRec *list_m = r.rowlist + (size_t)m * r.nt * ROW_CAP;
    int my_tile = -1, my_y = 0;
    uint32_t my_cnt = 0;
    uint32_t my_a = 0, my_b = 0;                   // hits per column octant of the tile, one BYTE each (octants 0-3 | 4-7; <= 64 hits per append)
    int k = 0;
    // Round 5: the kernel is bound by its SCALAR instructions (PMC: 590 SALU + 474 VALU per wave, one scalar unit per CU -> 1.0 ms of its
    // 1.25): the per-append histogram is kept byte-packed in two dwords while it is built and parked, and widened to the 16-bit fields
    // of the count words -- together with the octant mask of the list record -- by the 64 lanes of the flush at once.
    auto flush = [&]() {
        if (my_tile >= 0) {
            unsigned long long *w = cnt_m + 4 * (size_t)my_tile;
            const unsigned long long old = atomicAdd(w, 1ull | ((unsigned long long)my_cnt << 32));
            const unsigned long long lo = (unsigned long long)__builtin_amdgcn_perm(0u, my_a, 0x0c010c00u) | ((unsigned long long)__builtin_amdgcn_perm(0u, my_a, 0x0c030c02u) << 32);
            const unsigned long long hi = (unsigned long long)__builtin_amdgcn_perm(0u, my_b, 0x0c010c00u) | ((unsigned long long)__builtin_amdgcn_perm(0u, my_b, 0x0c030c02u) << 32);
            if (my_a) atomicAdd(w + 1, lo);                      // (no return value: fire and forget)
            if (my_b) atomicAdd(w + 2, hi);
            // octants with at least one hit: non-zero bytes (<= 64 each: + 0x7f sets bit 7 without a carry) -> bits 0-3 | 4-7
            const uint32_t na = (((my_a + 0x7f7f7f7fu) & 0x80808080u) >> 7) * 0x01020408u >> 24;
            const uint32_t nb = (((my_b + 0x7f7f7f7fu) & 0x80808080u) >> 7) * 0x01020408u >> 24;
            const uint32_t rm = (na & 0xfu) | ((nb & 0xfu) << 4);
            const uint32_t slot = (uint32_t)old;
            if (slot < (uint32_t)ROW_CAP) list_m[(size_t)my_tile * ROW_CAP + slot] = RowRec{(uint32_t)my_y | (rm << 24), ((uint32_t)stx << 8) | my_cnt};
        }
        my_tile = -1;
        k = 0;
    };
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int y = y_base + rr * TILE_H;
        if (y >= H) break;                                       // (wave-uniform)
        int t0 = -1, t1 = -1, t2 = -1, t3 = -1;                  // the <= 4 tiles this pixel's footprint touches
        uint32_t cm_a = 0, cm_b = 0;                             // column octants (bits) it touches in the left / right of them
        if (x < W) {
            const BinFoot f = bin_footprint(fx[rr], fy[rr], x, y, H, W, tiles_x);
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
            // exact hits per column octant, byte-packed: 8 ballots for a full append; the one-column overlaps into a neighbouring tile
            // (fewer than 8 hits: half of all appends) walk their <= 7 lanes with scalar operations instead
            uint32_t ha = 0, hb = 0;
            if (c >= 8u) {                                   // (wave-uniform)
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const uint32_t co = (uint32_t)__popcll(__ballot((lm >> o) & 1u));
                    if (o < 4) ha |= co << (8 * o); else hb |= co << (8 * (o - 4));
                }
            } else {
                for (unsigned long long mk = hmask; mk; mk &= mk - 1ull) {
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lm, __ffsll((long long)mk) - 1);
                    // bit o of l -> + 1 in the byte of octant o: bit i of a nibble times (1 + 2^7 + 2^14 + 2^21) lands on bit 8 i (and on
                    // bits that are masked away; no two products share a position)
                    ha += ((l & 0xfu) * 0x00204081u) & 0x01010101u;
                    hb += ((l >> 4) * 0x00204081u) & 0x01010101u;
                }
            }
            t0 = e0 ? -1 : t0;
            t1 = e1 ? -1 : t1;
            t2 = e2 ? -1 : t2;
            t3 = e3 ? -1 : t3;
            // round k's append is parked in lane k (all five values are wave-uniform: one v_writelane each, splat_types.hpp)
            {
                int v_cnt = (int)my_cnt, v_a = (int)my_a, v_b = (int)my_b;
                write_lane5(k, my_tile, T, v_cnt, (int)c, my_y, y, v_a, (int)ha, v_b, (int)hb);
                my_cnt = (uint32_t)v_cnt; my_a = (uint32_t)v_a; my_b = (uint32_t)v_b;
            }
            if (++k == 64) flush();
        }
    }
    flush();
    // (measured and rejected: the appends of a workgroup summed per tile in an LDS table first, one set of global atomics per distinct
    //  tile and workgroup -- 8x fewer atomics, 1027 -> 1167 us per clip: the kernel is bound by its VALU work, not by the atomics)
}

// One wave per (map, tile): the tile's row-segment list into image order (the appends arrived in any order; image order is what the
// staging loads and the record lists like best: unsorted +6..11 % on the one-flow operator), each segment with its first entry slot
// (exclusive prefix of the hit counts: where its hits go when the whole tile is one piece).  Done once per clip instead of by every
// workgroup that works on the tile (3.8 us of a 40 us workgroup life went into item -> lists -> sort -> scan).
// Record after the sort: image row | octants << 24, (x / 64) << 18 | first slot << 7 | hits.
// (cnt: the tiles' count words -- map m, tile t at cnt[m * cnt_map_stride + t * cnt_tile_stride], row segments in the low 32 bits;
//  src / dst: the lists, map m at + m * nt * ROW_CAP records; dst may be src.)
__global__ __launch_bounds__(CT) void rows_sort_kernel(const unsigned long long *__restrict__ cnt, size_t cnt_map_stride, uint32_t cnt_tile_stride,
                                                       const RowRec *src, RowRec *dst, uint32_t nt) {
    __shared__ uint32_t k_sy[CT / 64][ROW_CAP], k_sx[CT / 64][ROW_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * (CT / 64) + w, m = blockIdx.y;
    if (t >= nt) return;                                              // (whole waves; no barriers below)
    const uint32_t n = min((uint32_t)cnt[(size_t)m * cnt_map_stride + (size_t)t * cnt_tile_stride], (uint32_t)ROW_CAP);
    const RowRec *list_in = src + ((size_t)m * nt + t) * ROW_CAP;
    RowRec *list = dst + ((size_t)m * nt + t) * ROW_CAP;
    constexpr int PER = ROW_CAP / 64;
    RowRec rec[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t q = (uint32_t)lane + 64u * i;
        rec[i] = q < n ? list_in[q] : RowRec{0xffffffffu, 0xffffffffu};
        k_sy[w][q] = rec[i].sy & 0xffffffu;
        k_sx[w][q] = rec[i].sx_cnt >> 8;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t rank[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) rank[i] = 0;
    if (n <= 64u) {                                                      // (wave-uniform; the usual list: 20-60 segments -- one record per lane)
        const unsigned long long k0 = ((unsigned long long)(rec[0].sy & 0xffffffu) << 24) | (rec[0].sx_cnt >> 8);
        for (uint32_t j = 0; j < n; ++j) {
            const unsigned long long kj = ((unsigned long long)k_sy[w][j] << 24) | k_sx[w][j];
            rank[0] += kj < k0 ? 1u : 0u;                              // (keys are distinct: one append per (segment, tile))
        }
    } else {
        for (uint32_t j = 0; j < n; ++j) {
            const unsigned long long kj = ((unsigned long long)k_sy[w][j] << 24) | k_sx[w][j];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const unsigned long long ki = ((unsigned long long)(rec[i].sy & 0xffffffu) << 24) | (rec[i].sx_cnt >> 8);
                rank[i] += kj < ki ? 1u : 0u;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // hit counts in sorted order -> exclusive prefix (lane l owns sorted positions PER * l .. PER * l + PER - 1)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t q = (uint32_t)lane + 64u * i;
        if (q < n) k_sx[w][rank[i]] = rec[i].sx_cnt & 0xffu;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t c[PER], mine = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const uint32_t q = (uint32_t)(PER * lane + i); c[i] = q < n ? k_sx[w][q] : 0u; mine += c[i]; }
    uint32_t ex = wave_incl_scan(mine, lane) - mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < PER; ++i) { k_sy[w][PER * lane + i] = ex; ex += c[i]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t q = (uint32_t)lane + 64u * i;
        if (q < n) {
            const uint32_t base = min(k_sy[w][rank[i]], 0x7ffu);       // (meaningful only when the tile holds <= SEG entries)
            list[rank[i]] = RowRec{rec[i].sy, ((rec[i].sx_cnt >> 8) << ROWW_STX) | (base << ROWW_BASE) | (rec[i].sx_cnt & 0xffu)};
        }
    }
}

// One workgroup per frame: the (entries, row segments, octant histogram) words of its two maps -> work items in row-major tile
// order, the pieces of a heavy tile next to each other.  A piece = a range of the tile's 8 column octants (8 output columns each),
// cut greedily so that no piece's octant counts add up to more than SEG (an entry on an octant boundary counts in both octants: the
// sum bounds the piece from above) and the pieces weigh about the same.  Columns, not rows: a footprint is two pixels wide and two
// high, so 8 pieces by rows stage 1.78x the tile's entries, by columns 1.10x.  An octant that holds more than SEG entries by itself
// makes a piece that its workgroup finds too long and hands to the pass-by-pass launch.
struct ClipPlan {
    ItemDesc *items;               // [nframes][items_cap]
    uint32_t *totals;              // [nframes][CLIP_TOTALS]
    uint32_t items_cap;
};

// (words: frame fi's two maps at words0 + fi * frame_stride and words1 + fi * frame_stride, tile t at + t * tile_stride.  APPROX: the
//  words of the one-flow binning (slr_splat_bin: one histogram word, 8 bits per octant in units of 16 entries, small appends left out)
//  -- the cut is then an estimate: pieces of at most 7/8 of a segment by the scaled histogram, and a piece that still turns out too
//  long goes to the pass-by-pass launch.)
template <bool APPROX>
__global__ __launch_bounds__(CT) void rows_plan_pair_kernel(const unsigned long long *__restrict__ words0, const unsigned long long *__restrict__ words1,
                                                            size_t frame_stride, uint32_t tile_stride, uint32_t nt, ClipPlan p, uint32_t seg) {
    __shared__ uint32_t wsum[CT / 64];
    const uint32_t fi = blockIdx.x;
    const int tid = threadIdx.x;
    const unsigned long long *w0 = words0 + (size_t)fi * frame_stride, *w1 = words1 + (size_t)fi * frame_stride;
    ItemDesc *items = p.items + (size_t)fi * p.items_cap;
    uint32_t run = 0;
    // Heavy tiles first (two passes over the same row-major order): the frames of a batch are interleaved, so the launch ends where
    // all its frames end, and a ridge tile (long record lists: up to 90 us against a mean of 37) that starts there keeps the chip
    // waiting.  Heavy = cut into pieces, or more than SLR_CLIP_HEAVY / 8 of a segment.
    const uint32_t heavy_thr = (seg * (uint32_t)SLR_CLIP_HEAVY) / 8u;
    for (uint32_t pb = 0; pb < (SLR_CLIP_HEAVY ? 2u : 1u) * ((nt + CT - 1) / CT) * CT; pb += CT) {
        const uint32_t pass = pb / (((nt + CT - 1) / CT) * CT), b = pb - pass * (((nt + CT - 1) / CT) * CT);
        const uint32_t t = b + tid;
        bool on = t < nt;
        unsigned long long a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (on) { const size_t q = (size_t)t * tile_stride;
                  a0 = w0[q]; a1 = w0[q + 1]; a2 = APPROX ? 0ull : w0[q + 2];
                  b0 = w1[q]; b1 = w1[q + 1]; b2 = APPROX ? 0ull : w1[q + 2]; }
        const uint32_t cnt = (uint32_t)(a0 >> 32) + (uint32_t)(b0 >> 32);
        if (SLR_CLIP_HEAVY && on && (cnt > heavy_thr) != (pass == 0u)) on = false;      // not this pass's tile
        unsigned long long pcs = 0x80ull;                          // pieces: (first octant | octants << 4), 8 bits each; default: octants [0, 8)
        uint32_t ns = on ? 1u : 0u;
        if (on && cnt > seg) {
            uint32_t oh[8], osum = 0;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                if (APPROX) oh[o] = (uint32_t)((a1 >> (8 * o)) & 0xffu) + (uint32_t)((b1 >> (8 * o)) & 0xffu);
                else {
                    const unsigned long long x = o < 4 ? a1 : a2, y = o < 4 ? b1 : b2;
                    oh[o] = (uint32_t)((x >> (16 * (o & 3))) & 0xffffu) + (uint32_t)((y >> (16 * (o & 3))) & 0xffffu);
                }
                osum += oh[o];
            }
            uint32_t limit = seg;
            if (APPROX) {                                  // scale the estimate to the tile's count (+ 1/8 for entries on octant boundaries)
                const float scale = (float)(cnt + cnt / 8u) / (float)(osum ? osum : 1u);
                osum = 0;
#pragma unroll
                for (int o = 0; o < 8; ++o) { oh[o] = (uint32_t)((float)oh[o] * scale); osum += oh[o]; }
                limit = (seg * 7u) / 8u;
                osum = osum ? osum : 1u;
            }
            const uint32_t even = osum / ((osum + limit - 1u) / limit);             // pieces of about equal weight, not one full + a rest
            uint32_t start = 0, sum = 0, np = 0;
            unsigned long long q = 0;
#pragma unroll
            for (uint32_t o = 0; o < 8; ++o) {
                if ((sum + oh[o] > limit || sum + oh[o] / 2u >= even) && o > start) { q |= (unsigned long long)(start | ((o - start) << 4)) << (8 * np); ++np; start = o; sum = 0; }
                sum += oh[o];
            }
            q |= (unsigned long long)(start | ((8u - start) << 4)) << (8 * np); ++np;
            pcs = q; ns = np;
        }
        // SLR_CLIP_ALIGNED: the first piece of tile t is item t in EVERY frame, further pieces follow behind item nt - 1: item i of the
        // frames of an interleaved batch is then the same tile, their workgroups run side by side on one XCD and fetch the source
        // region once (FETCH_SIZE per frame 514 -> ... MB); in line, a frame's items drift against its neighbours' by the extra pieces.
        const uint32_t ex = block_excl_scan(SLR_CLIP_ALIGNED ? (ns ? ns - 1u : 0u) : ns, wsum, tid);
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < CT / 64; ++w) tot += wsum[w];
        __syncthreads();
        if (on) {
            ItemDesc dsc;
            dsc.tile = t; dsc.cnt0 = (uint32_t)(a0 >> 32); dsc.cnt1 = (uint32_t)(b0 >> 32);
            dsc.off0 = (uint32_t)a0; dsc.off1 = (uint32_t)b0;                           // row segments appended for the two directions
            dsc.partoff = 0;
            for (uint32_t k = 0; k < ns; ++k) {
                const uint32_t pc = (uint32_t)(pcs >> (8 * k)) & 0xffu;
                dsc.seg = pc & 0xfu;                                                    // first column octant of the piece
                dsc.nseg = pc >> 4;                                                     // its octants (8 = the whole tile)
                const uint32_t at = !SLR_CLIP_ALIGNED ? run + ex + k : k == 0 ? t : nt + run + ex + k - 1u;
                if (at < p.items_cap) items[at] = dsc;
            }
        }
        run += tot;
    }
    if (SLR_CLIP_ALIGNED) run += nt;
    if (tid == 0) {
        uint32_t *tt = p.totals + (size_t)fi * CLIP_TOTALS;
        tt[0] = run < p.items_cap ? run : p.items_cap; tt[1] = 0; tt[2] = 0; tt[3] = 0; tt[4] = 0; tt[5] = 0; tt[6] = 0; tt[7] = 0;
    }
}

// =========================================================================== per batch of frames: the fused tile kernel

// grid: per frame a multiple of 8 * C_XCD blocks, the frames' groups interleaved (see launch); CT work-items; ClipCfg::LDS_BYTES of LDS.
// PASSES = false: one piece per workgroup, no loops over work; a piece of more than SEG entries goes to the frame's deferred list.
// PASSES = true:  C_DEFER_WG workgroups per frame walk the deferred lists pass by pass.
// (Round 4, measured and rejected: as many workgroups as the chip holds, each walking virtual blocks w, w + grid, ... -- a slot idles ~7 us
//  between two workgroups of ~75 us.  The loop keeps the by-value kernel arguments live around the whole body: 80 spilled SGPRs, and the
//  kernel went from 152 to 190 us per frame of work even with one round per workgroup, 202 us with 512 persistent ones.  It needs the
//  per-frame arguments in memory, read per item, first.)
template <bool G2, bool PASSES, bool B4 = false>
__global__ __launch_bounds__(CT, PASSES ? 1 : SLR_WAVES_CLIP) void clip_tile_kernel(ClipBatch b) {
    using Cfg = std::conditional_t<PASSES, std::conditional_t<B4, ClipPassCfgB4, ClipPassCfg>, std::conditional_t<B4, ClipCfgB4, ClipCfg>>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const TileLds<Cfg> L(smem);
    const TileShared &s = b.s;
    uint32_t bf, bx;
    if (PASSES) { bf = blockIdx.x / C_DEFER_WG; bx = blockIdx.x % C_DEFER_WG; }
    else if (b.interleave) {
        constexpr uint32_t G = 8 * C_XCD;
        const uint32_t gg = blockIdx.x / G;
        bf = gg % b.nb;
        bx = (gg / b.nb) * G + blockIdx.x % G;
    } else {
        bf = 0; bx = blockIdx.x;
#pragma unroll
        for (int i = 0; i + 1 < C_MAXB; ++i)
            if (i + 1 < (int)b.nb && bx >= b.f[i].grid) { bx -= b.f[i].grid; bf = i + 1; }
    }
    const TileFrame &f = b.f[bf];
    const int tid = threadIdx.x;
    if (PASSES && f.totals[4] == 0u) return;               // the normal case: nothing deferred in this frame, nothing to reset
    const TileScalars k = tile_scalars(s, f);
    if (!PASSES) {
        if (bx >= f.grid) return;                          // this frame has fewer groups than the longest of the batch
        const uint32_t item = xcd_item(bx);
        if (item >= f.totals[0]) return;
        const Piece p = make_piece<Cfg>(s, f.items[item]);
        if (!rows_piece_once<Cfg, true, true, false, G2>(s, f, L, p, tid, k, 0, s.C) && tid == 0)
            f.defer[atomicAdd(f.totals + 4, 1u)] = item;   // more than one pass: the pass-by-pass launch takes the piece
    } else {
        const uint32_t ndef = f.totals[4];
        for (uint32_t q = bx; q < ndef; q += C_DEFER_WG)
            rows_piece_passes<Cfg, true, true, false, G2>(s, f, L, make_piece<Cfg>(s, f.items[f.defer[q]]), tid, k, 0, s.C);
        // the last workgroup of the frame to get here empties the deferred list for the plan's next use (everybody has read it)
        __syncthreads();
        if (tid == 0 && atomicAdd(f.totals + 5, 1u) == C_DEFER_WG - 1u) { f.totals[4] = 0u; f.totals[5] = 0u; }
    }
}

// =========================================================================== host side

struct ClipLayout {
    int tiles_x, tiles_y, tiles;
    uint32_t nt, nframes, nmaps, items_cap;
    size_t off_rowcnt, off_rowlist, off_items, off_totals, off_defer, total;
};

static ClipLayout clip_layout(int nframes, int H, int W) {
    ClipLayout L;
    L.tiles_x = (W + TILE_W - 1) / TILE_W;
    L.tiles_y = (H + TILE_H - 1) / TILE_H;
    L.tiles = L.tiles_x * L.tiles_y;
    L.nt = (uint32_t)L.tiles;
    L.nframes = (uint32_t)nframes;
    L.nmaps = 2u * L.nframes;
    L.items_cap = 8u * L.nt;                           // a tile has at most 8 pieces
    size_t o = 0;
    L.off_rowcnt = o;  o += al256((size_t)L.nmaps * L.nt * 32);
    L.off_totals = o;  o += al256((size_t)L.nframes * CLIP_TOTALS * 4);
    L.off_items = o;   o += al256((size_t)L.nframes * L.items_cap * sizeof(ItemDesc));
    L.off_defer = o;   o += al256((size_t)L.nframes * L.items_cap * 4);
    L.off_rowlist = o; o += al256((size_t)L.nmaps * L.nt * ROW_CAP * sizeof(RowRec));
    L.total = o;
    return L;
}

static int clip_check(int nframes, int C, int H, int W, const char *who) {
    // image rows travel in 24 bits of a row-list entry; 8 planes fit one buffer descriptor (a larger plane stack: several launches, plane_group)
    if (nframes <= 0 || nframes > 16384 || H <= 0 || W <= 0 || H >= (1 << 24) || (long long)H * W >= (1LL << 26)) {
        set_error("%s: bad sizes nframes=%d C=%d H=%d W=%d (H*W < 2^26)", who, nframes, C, H, W);
        return SLR_E_BADARG;
    }
    return 0;
}

template <bool G2, bool PASSES, bool B4 = false>
static int launch_clip_kernel(const ClipBatch &b, uint32_t grid, hipStream_t st) {
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device
    static LdsOptIn attr;
    if (int e = lds_opt_in((const void *)clip_tile_kernel<G2, PASSES, B4>, 159 * 1024, attr)) return e;
    hipLaunchKernelGGL((clip_tile_kernel<G2, PASSES, B4>), dim3(grid), dim3(CT), PASSES ? ClipPassCfg::LDS_BYTES : ClipCfg::LDS_BYTES, st, b);
    return 0;
}
template <bool PASSES>
static int launch_clip_kernel(const ClipBatch &b, uint32_t grid, bool g2, bool b4, hipStream_t st) {
    if (g2) return b4 ? launch_clip_kernel<true, PASSES, true>(b, grid, st) : launch_clip_kernel<true, PASSES, false>(b, grid, st);
    return b4 ? launch_clip_kernel<false, PASSES, true>(b, grid, st) : launch_clip_kernel<false, PASSES, false>(b, grid, st);
}

extern thread_local void *g_ev_start, *g_ev_stop;          // slr_splat_time_next (splat.hip)
#ifdef SLR_TRACE
extern long long *g_trace;                                 // slr_debug_trace (splat.hip)
#endif

}  // namespace slr

using namespace slr;

SLR_EXPORT size_t slr_clip_plan_bytes(int nframes, int H, int W) {
    if (nframes <= 0 || nframes > 16384 || H <= 0 || W <= 0 || H >= (1 << 24) || (long long)H * W >= (1LL << 26)) return 0;
    return clip_layout(nframes, H, W).total;
}

SLR_EXPORT int slr_clip_plan_totals(int nframes, int H, int W, size_t *offset_bytes, int *stride_words) {
    if (int e = clip_check(nframes, 1, H, W, __func__)) return e;
    SLR_CHECK_ARG(offset_bytes && stride_words, "null pointer");
    *offset_bytes = clip_layout(nframes, H, W).off_totals;
    *stride_words = (int)CLIP_TOTALS;
    return 0;
}

SLR_EXPORT int slr_clip_plan_build(const float *disp_f, const int *idx_f, const float *disp_p, const int *idx_p, int nframes,
                                   int H, int W, void *plan, size_t plan_bytes, void *stream) {
    SLR_CHECK_ARG(disp_f && idx_f && disp_p && idx_p && plan, "null pointer");
    if (int e = clip_check(nframes, 1, H, W, __func__)) return e;
    const ClipLayout L = clip_layout(nframes, H, W);
    if (((uintptr_t)plan & 15) || plan_bytes < L.total) {
        set_error("%s: plan buffer needs %zu bytes (16-byte aligned), got %zu", __func__, L.total, plan_bytes);
        return SLR_E_WORKSPACE;
    }
    char *b = (char *)plan;
    hipStream_t st = (hipStream_t)stream;
    ClipRows r = {};
    r.disp[0] = disp_f; r.disp[1] = disp_p; r.idx[0] = idx_f; r.idx[1] = idx_p;
    r.rowcnt = (unsigned long long *)(b + L.off_rowcnt);
    r.rowlist = (RowRec *)(b + L.off_rowlist);
    r.nframes = L.nframes; r.nt = L.nt;
    ClipPlan p = {};
    p.items = (ItemDesc *)(b + L.off_items);
    p.totals = (uint32_t *)(b + L.off_totals);
    p.items_cap = L.items_cap;
    const size_t nwords = (size_t)L.nmaps * L.nt * 4;
    hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, r.rowcnt, nwords);
    const dim3 grid((unsigned)(L.tiles_x * ((L.tiles_y + SLR_ROWBIN_CLIP_R - 1) / SLR_ROWBIN_CLIP_R)), L.nmaps);      // (blockIdx.y carries the map: <= 32768 maps, see clip_check)
    hipLaunchKernelGGL(rowbin_clip_kernel, grid, dim3(CT), 0, st, r, H, W, L.tiles_x, L.tiles_y);
    hipLaunchKernelGGL(rows_sort_kernel, dim3((L.nt + CT / 64 - 1) / (CT / 64), L.nmaps), dim3(CT), 0, st, (const unsigned long long *)r.rowcnt,
                       (size_t)L.nt * 4, 4u, (const RowRec *)r.rowlist, r.rowlist, L.nt);
    hipLaunchKernelGGL(rows_plan_pair_kernel<false>, dim3(L.nframes), dim3(CT), 0, st, (const unsigned long long *)r.rowcnt,
                       (const unsigned long long *)r.rowcnt + (size_t)L.nframes * L.nt * 4, (size_t)L.nt * 4, 4u, L.nt, p, (uint32_t)C_SEG);
    SLR_CHECK_LAUNCH();
    return 0;
}

// The fused kernel (+ the pass-by-pass launch) over the frames of `b`, once per plane group (plane_group, splat_op.hip: one group unless
// the plane stack reaches 2 GiB).  Groups after the first take the kernel without the second weight group and write no normaliser.
static int launch_clip_groups(ClipBatch &b, uint32_t grid, bool g2, hipStream_t st, bool b4 = false) {
    const int C = b.s.C, gp = plane_group(C, b.s.H, b.s.W);
    SLR_CHECK_ARG(!b4 || (C % 4 == 0 && gp >= C), "plane-blocked values: C % 4 == 0 and the whole stack below 2 GiB");
    const size_t hw = (size_t)b.s.H * b.s.W;
    const float *values = b.s.in;
    float *outs[C_MAXB];
    for (uint32_t k = 0; k < b.nb; ++k) outs[k] = b.f[k].out;
    b.s.Cs = C;
    for (int pb = 0; pb < C; pb += gp) {
        b.s.in = values + (size_t)pb * hw;
        b.s.C = C - pb < gp ? C - pb : gp;
        for (uint32_t k = 0; k < b.nb; ++k) {
            b.f[k].out = outs[k] + (size_t)pb * hw;
            if (pb) { b.f[k].norm_out = nullptr; b.f[k].out2 = nullptr; }
        }
        const bool two = g2 && pb == 0;
        if (pb == 0 && g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));        // slr_splat_time_next: the dominant kernel only
        if (grid) {
            if (int e = launch_clip_kernel<false>(b, grid, two, b4, st)) return e;
        }
        if (pb == 0 && g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
        if (pb == 0) g_ev_start = g_ev_stop = nullptr;
        // pieces of more than SEG entries (none for ordinary flows): pass by pass
        if (int e = launch_clip_kernel<true>(b, b.nb * C_DEFER_WG, two, b4, st)) return e;
    }
    SLR_CHECK_LAUNCH();
    return 0;
}

static int synth_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                            const float *values2, const float *wlogit2, int exp_weights2, float *const *out2,
                            const float *const *disp_f, const float *const *disp_p, const float *alpha,
                            float *const *out, float *const *norm_out, int C, int H, int W, float eps,
                            void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                            const int *n_items, void *stream) {
    SLR_CHECK_ARG(values && wlogit && disp_f && disp_p && alpha && out && plan && frame, "null pointer");
    SLR_CHECK_ARG((!values2 && !wlogit2 && !out2) || (values2 && wlogit2 && out2), "the second group needs values, weights and outputs");
    SLR_CHECK_ARG(nb >= 1 && nb <= C_MAXB, "1 <= nb <= 16 frames per launch");
    SLR_CHECK_ARG(C >= 1, "C");
    if (int e = clip_check(nframes, C, H, W, __func__)) return e;
    const ClipLayout L = clip_layout(nframes, H, W);
    if (((uintptr_t)plan & 15) || plan_bytes < L.total) {
        set_error("%s: plan needs %zu bytes (got %zu), 16-byte aligned", __func__, L.total, plan_bytes);
        return SLR_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *pb = (char *)plan;
    ClipBatch b = {};
    b.s.in = values; b.s.mul = wlogit; b.s.mulmax = wmax; b.s.in2 = values2; b.s.mul2 = wlogit2;
    b.s.N = 1; b.s.C = C; b.s.H = H; b.s.W = W; b.s.tiles_x = L.tiles_x; b.s.tiles = L.tiles;
    const bool b4 = (exp_weights & SLR_SYNTH_VALUES_B4) != 0;
    b.s.mulmode = wmax ? MUL_EXP_SHIFT : ((exp_weights & 1) ? MUL_EXP : MUL_PLANE);
    b.s.mulmode2 = exp_weights2 ? MUL_EXP : MUL_PLANE;
    b.s.norm_mode = SLR_NORM_CLAMP_EPS;
    b.s.eps = eps;
#ifdef SLR_TRACE
    b.s.trace = g_trace;
#endif
    b.nb = (uint32_t)nb;
    b.interleave = (SLR_BATCH_INTERLEAVE && nb > 1) ? 1u : 0u;
    uint32_t gsum = 0, gmax = 0;
    for (int k = 0; k < nb; ++k) {
        SLR_CHECK_ARG(frame[k] >= 0 && frame[k] < nframes, "frame index");
        for (int j = 0; j < k; ++j) SLR_CHECK_ARG(frame[j] != frame[k], "a frame twice in one batch (its deferred list would be shared)");
        SLR_CHECK_ARG(disp_f[k] && disp_p[k] && out[k], "null pointer");
        TileFrame &f = b.f[k];
        const size_t i = (size_t)frame[k];
        f.flow[0] = disp_f[k]; f.flow[1] = disp_p[k];
        f.rowlist[0] = (const RowRec *)(pb + L.off_rowlist) + i * L.nt * ROW_CAP;
        f.rowlist[1] = (const RowRec *)(pb + L.off_rowlist) + ((size_t)L.nframes + i) * L.nt * ROW_CAP;
        f.items = (const ItemDesc *)(pb + L.off_items) + i * L.items_cap;
        f.totals = (uint32_t *)(pb + L.off_totals) + i * CLIP_TOTALS;
        f.defer = (uint32_t *)(pb + L.off_defer) + i * L.items_cap;
        f.items_cap = L.items_cap;
        f.out = out[k]; f.norm_out = norm_out ? norm_out[k] : nullptr;
        if (values2) { SLR_CHECK_ARG(out2[k], "null pointer"); f.out2 = out2[k]; }
        f.scale[0] = alpha[k]; f.scale[1] = 1.0f - alpha[k];
        // what the host knows of the plan (read back once per clip); unknown: the grid covers the bound, surplus workgroups exit at once
        const int ni = n_items ? n_items[k] : -1;
        const uint32_t cover = ni >= 0 && (uint32_t)ni < L.items_cap ? (uint32_t)ni : L.items_cap;
        f.grid = ((cover + 8 * C_XCD - 1) / (8 * C_XCD)) * 8 * C_XCD;
        gsum += f.grid;
        gmax = f.grid > gmax ? f.grid : gmax;
    }
    // Frames of a batch are consecutive frames of a clip: tile T of frame k+1 gathers from almost the same source region as tile T
    // of frame k.  With the groups of 8 * C_XCD blocks dealt round-robin over the frames they run side by side on the same XCD and
    // share its L2 (frames with fewer groups leave a few empty blocks).
    const uint32_t grid = b.interleave ? gmax * b.nb : gsum;
    return launch_clip_groups(b, grid, values2 != nullptr, st, b4);
}

// slr_synth_group: one frame from two workspaces that slr_splat_bin / slr_splat_bin_pair filled (no clip plan).  Their row lists are
// copied in image order with first slots into the forward workspace's second list area, a two-flow plan is made from the two
// workspaces' per-tile words (estimated histograms), then the fused kernel of the clip path runs on that one frame.
SLR_EXPORT int slr_synth_group(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                               const float *disp_f, const float *disp_p, float alpha, float *out,
                               float *norm_out, int C, int H, int W, float eps, void *ws_f, void *ws_p,
                               size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(values && wlogit && disp_f && disp_p && out, "null pointer");
    SLR_CHECK_ARG(ws_f != ws_p, "the two flows need separate workspaces");
    if (int e = clip_check(1, C, H, W, __func__)) return e;
    OpWs wf, wp;
    if (int e = op_ws_open(wf, 1, H, W, ws_f, ws_bytes, __func__)) return e;
    if (int e = op_ws_open(wp, 1, H, W, ws_p, ws_bytes, __func__)) return e;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nt = wf.L.nt;
    const dim3 sgrid((nt + CT / 64 - 1) / (CT / 64), 1);
    hipLaunchKernelGGL(rows_sort_kernel, sgrid, dim3(CT), 0, st, (const unsigned long long *)wf.rowinfo, (size_t)0, 2u, (const RowRec *)wf.rowlist, wf.rowlist2, nt);
    hipLaunchKernelGGL(rows_sort_kernel, sgrid, dim3(CT), 0, st, (const unsigned long long *)wp.rowinfo, (size_t)0, 2u, (const RowRec *)wp.rowlist,
                       wf.rowlist2 + (size_t)nt * ROW_CAP, nt);
    ClipPlan p = {};
    p.items = wf.items2; p.totals = wf.totals + 8; p.items_cap = wf.L.items2_cap;
    hipLaunchKernelGGL(rows_plan_pair_kernel<true>, dim3(1), dim3(CT), 0, st, (const unsigned long long *)wf.rowinfo, (const unsigned long long *)wp.rowinfo,
                       (size_t)0, 2u, nt, p, (uint32_t)C_SEG);
    ClipBatch b = {};
    b.s.in = values; b.s.mul = wlogit; b.s.mulmax = wmax;
    b.s.N = 1; b.s.C = C; b.s.H = H; b.s.W = W; b.s.tiles_x = wf.L.tiles_x; b.s.tiles = wf.L.tiles;
    b.s.mulmode = wmax ? MUL_EXP_SHIFT : (exp_weights ? MUL_EXP : MUL_PLANE);
    b.s.norm_mode = SLR_NORM_CLAMP_EPS;
    b.s.eps = eps;
#ifdef SLR_TRACE
    b.s.trace = g_trace;
#endif
    b.nb = 1; b.interleave = 0;
    TileFrame &f = b.f[0];
    f.flow[0] = disp_f; f.flow[1] = disp_p;
    f.rowlist[0] = wf.rowlist2; f.rowlist[1] = wf.rowlist2 + (size_t)nt * ROW_CAP;
    f.items = wf.items2; f.totals = wf.totals + 8; f.defer = wf.defer2; f.items_cap = wf.L.items2_cap;
    f.out = out; f.norm_out = norm_out;
    f.scale[0] = alpha; f.scale[1] = 1.0f - alpha;
    f.grid = ((wf.L.items2_cap + 8 * C_XCD - 1) / (8 * C_XCD)) * 8 * C_XCD;
    return launch_clip_groups(b, f.grid, false, st);
}

// [N,C,H,W] -> [N,C/4,H,W,4]: what SLR_SYNTH_VALUES_B4 reads (a clip's feature planes, once per clip)
__global__ __launch_bounds__(256) void pack_planes4_kernel(const float *__restrict__ in, float4 *__restrict__ out, size_t hw, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t q = i / hw, p = i - q * hw;                   // q = sample * C/4 + chunk
        const float *src = in + q * 4 * hw + p;
        out[i] = make_float4(src[0], src[hw], src[2 * hw], src[3 * hw]);
    }
}

SLR_EXPORT int slr_pack_planes4(const float *in, float *out, int N, int C, int H, int W, void *stream) {
    SLR_CHECK_ARG(in && out && in != out, "null pointer / in place");
    SLR_CHECK_ARG(N > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0 && !((uintptr_t)out & 15), "N, C % 4 == 0, H, W; 16-byte aligned output");
    const size_t hw = (size_t)H * W, total = (size_t)N * (C / 4) * hw;
    const unsigned grid = (unsigned)((total + 255) / 256 < 65536u * 16u ? (total + 255) / 256 : 65536u * 16u);
    hipLaunchKernelGGL(pack_planes4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (float4 *)out, hw, total);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_synth_group_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                          const float *const *disp_f, const float *const *disp_p, const float *alpha,
                                          float *const *out, float *const *norm_out, int C, int H, int W, float eps,
                                          void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                                          const int *n_items, void *stream) {
    return synth_clip_batch(values, wlogit, wmax, exp_weights, nullptr, nullptr, 0, nullptr, disp_f, disp_p, alpha, out, norm_out,
                            C, H, W, eps, plan, plan_bytes, nframes, frame, nb, n_items, stream);
}

SLR_EXPORT int slr_synth_two_groups_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                               const float *values2, const float *wlogit2, int exp_weights2,
                                               const float *const *disp_f, const float *const *disp_p, const float *alpha,
                                               float *const *out, float *const *out2, int C, int H, int W, float eps,
                                               void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                                               const int *n_items, void *stream) {
    SLR_CHECK_ARG(values2 && wlogit2 && out2, "null pointer");
    return synth_clip_batch(values, wlogit, wmax, exp_weights, values2, wlogit2, exp_weights2, out2, disp_f, disp_p, alpha, out,
                            nullptr, C, H, W, eps, plan, plan_bytes, nframes, frame, nb, n_items, stream);
}

SLR_EXPORT int slr_synth_group_clip(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                    const float *disp_f, const float *disp_p, float alpha, float *out, float *norm_out,
                                    int C, int H, int W, float eps, void *plan, size_t plan_bytes, int nframes,
                                    int frame, int n_items, void *stream) {
    return slr_synth_group_clip_batch(values, wlogit, wmax, exp_weights, &disp_f, &disp_p, &alpha, &out,
                                      norm_out ? &norm_out : nullptr, C, H, W, eps, plan, plan_bytes, nframes, &frame, 1,
                                      &n_items, stream);
}
