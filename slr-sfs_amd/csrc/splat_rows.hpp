// splat_rows.hpp -- the ROWS front end of the tile kernels: a piece's entries from the row-segment lists of its tile.
// A row segment = 64 consecutive source pixels of one image row (one wave's coalesced load).  A binning pass (rowbin kernels: per call
// in splat_op.hip, per clip in splat_clip.hip) has appended every segment to the few 8x64 OUTPUT tiles its footprints touch, with its
// exact number of hits; the tile kernel walks exactly the listed rows of the flow (coalesced), tests the footprints against its
// piece and writes the hits to the entry array.  Entry slots:
//   MODE 0  the whole tile, no overflow, <= SEG entries: the list's hit counts give every row segment its first slot;
//   MODE 1  a column range / an overflowed list: one LDS atomic per wave and row segment hands out the slots (L.misc[0]);
//   MODE 2  ordinals (hits of the waves before + own so far, after a count pass): emits the ordinals in [lo, hi) -- the pass-by-pass
//           walk of a piece that holds more than SEG entries.
#pragma once
#include "splat_tile.hpp"

namespace slr {

// list record after the sort: image row | column octants << 24, (x / 64) << 18 | first slot << 7 | hits
constexpr uint32_t ROWW_STX = 18, ROWW_BASE = 7;

template <class Cfg>
__device__ __forceinline__ Piece make_piece(const TileShared &s, const ItemDesc &it) {
    Piece p;
    p.tile = it.tile;
    p.n = (int)(it.tile / (uint32_t)s.tiles);
    const uint32_t tl = it.tile - (uint32_t)p.n * (uint32_t)s.tiles;
    p.ty0 = (int)(tl / (uint32_t)s.tiles_x) * TILE_H;
    p.tx0 = (int)(tl % (uint32_t)s.tiles_x) * TILE_W;
    const int noct = (int)min(max(it.nseg, 1u), 8u);
    p.pca = 8 * (int)min(it.seg, 7u);
    p.pcb = min(p.pca + 8 * noct, TILE_W);
    p.whole = p.pca == 0 && p.pcb == TILE_W;
    // (it.off0 / off1: row segments appended for the two directions; one flow: it.cnt1 carries them, rowbin_kernel's word)
    const uint32_t nr0 = Cfg::NDIR > 1 ? it.off0 : it.cnt1, nr1 = Cfg::NDIR > 1 ? it.off1 : 0u;
    p.cnt0 = it.cnt0; p.cnt1 = Cfg::NDIR > 1 ? it.cnt1 : 0u;
    p.ovf0 = nr0 > (uint32_t)ROW_CAP; p.ovf1 = nr1 > (uint32_t)ROW_CAP;
    const uint32_t all = (uint32_t)s.H * (uint32_t)s.tiles_x;
    p.len0 = p.ovf0 ? all : nr0; p.len1 = p.ovf1 ? all : nr1;
    p.n0 = p.ovf0 ? 0u : nr0;
    return p;
}

// The row-segment list(s) of the tile, already in image order with first slots (rows_sort_clip_kernel) -> LDS, compacted:
// direction 0, then direction 1.
template <class Cfg>
__device__ __forceinline__ void rows_setup_sorted(const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid) {
    static_assert(TT >= Cfg::NDIR * ROW_CAP, "one work-item per list slot");
    const int d = tid / ROW_CAP, q = tid & (ROW_CAP - 1);
    if (d < Cfg::NDIR) {
        const uint32_t nd = d ? (p.ovf1 ? 0u : p.len1) : p.n0;
        if ((uint32_t)q < nd) {
            const RowRec *list = d ? f.rowlist[Cfg::NDIR - 1] : f.rowlist[0];      // (a select, not an index: the frame's arguments may live in registers)
            const RowRec r = list[(size_t)p.tile * ROW_CAP + q];
            const uint32_t pos = (d ? p.n0 : 0u) + (uint32_t)q;
            L.rl[pos] = r.sy;
            L.rl[Cfg::NDIR * ROW_CAP + pos] = r.sx_cnt;
        }
    }
    __syncthreads();
}

// One flow, list as the binning kernel left it (appends in any order: image row | octants << 24, (x / 64) << 8 | hits): put into
// image order by ranking every key among the others (<= ROW_CAP broadcast reads), exclusive prefix of the hit counts -> the same LDS
// words as rows_setup_sorted.  (Image order is what the staging loads and the record lists like best: unsorted +6..11 %.)
template <class Cfg>
__device__ __forceinline__ void rows_setup_unsorted(const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid) {
    static_assert(Cfg::NDIR == 1, "the clip's lists are sorted per clip");
    uint32_t *ssy = L.rl, *sw1 = L.rl + ROW_CAP, *usy = L.rl + 2 * ROW_CAP, *usx = L.rl + 3 * ROW_CAP;
    const uint32_t nd = p.n0;
    RowRec r = {0u, 0u};
    if ((uint32_t)tid < nd) { r = f.rowlist[0][(size_t)p.tile * ROW_CAP + tid]; usy[tid] = r.sy; usx[tid] = r.sx_cnt >> 8; }
    __syncthreads();
    const unsigned long long mykey = ((unsigned long long)(r.sy & 0xffffffu) << 24) | (r.sx_cnt >> 8);
    uint32_t rank = 0;
    if ((uint32_t)tid < nd)
        for (uint32_t j = 0; j < nd; ++j) {
            const unsigned long long k2 = ((unsigned long long)(usy[j] & 0xffffffu) << 24) | usx[j];
            rank += (k2 < mykey) ? 1u : 0u;                            // (keys are distinct: one append per (segment, tile))
        }
    if ((uint32_t)tid < nd) { ssy[rank] = r.sy; sw1[rank] = ((r.sx_cnt >> 8) << ROWW_STX) | (r.sx_cnt & 0xffu); }
    __syncthreads();
    const uint32_t w1 = (uint32_t)tid < nd ? sw1[tid] : 0u;
    const uint32_t ex = block_excl_scan(w1 & 0xffu, L.wsum, tid);
    if ((uint32_t)tid < nd) sw1[tid] = w1 | (min(ex, 0x7ffu) << ROWW_BASE);
    __syncthreads();
}

// One walk over this wave's row segments (wave w takes segments w, w + 8, ... of [direction 0 ; direction 1]); returns the wave's hits.
// EMIT: write the entries (source pixel | direction << 31, target X, Y, weight logit [+ second group's logit and value]) with slots
// in [lo, hi) to the entry array.  The weight logits of a row segment are loaded with its flow: coalesced, and the dependent gather
// round trip they used to be, after the entries were known, is gone from phase 1.
// GLB: the entries go to the global array `gent` (at their slot) instead of the entry array in LDS -- a piece of more than a segment
// written out once for the sink launch of the scan front end (splat_op.hip).
template <class Cfg, int MODE, bool EMIT, bool G2, bool GLB = false>
__device__ __forceinline__ uint32_t rows_walk(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid,
                                              uint32_t wave_base, uint32_t lo, uint32_t hi, float4 *gent = nullptr) {
    static_assert(!GLB || !G2, "one weight group");
    constexpr int CB = Cfg::NDIR > 1 ? SLR_ROW_CB_CLIP : SLR_ROW_CB;
    const int lane = tid & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t HW = (uint32_t)(s.H * s.W);
    const uint32_t nseg = p.len0 + p.len1;
    const uint32_t my_n = nseg > wid ? (nseg - wid + (uint32_t)(TT / 64) - 1u) / (uint32_t)(TT / 64) : 0u;
    const uint32_t range_mask = ((1u << ((p.pcb - p.pca) >> 3)) - 1u) << (p.pca >> 3);      // the piece's column octants
    const uint32_t *rl_sy = L.rl, *rl_w1 = L.rl + Cfg::NDIR * ROW_CAP;
    struct Group { float fx[CB], fy[CB], z[CB], l2[G2 ? CB : 1], v2[G2 ? CB : 1]; int sy[CB], stx[CB]; uint32_t b0[CB], d[CB]; };
    const bool has_mul = s.mulmode != MUL_ONE && s.mul != nullptr;
    const float *fl0 = f.flow[0] + (size_t)p.n * 2 * HW, *fl1 = Cfg::NDIR > 1 ? f.flow[1] + (size_t)p.n * 2 * HW : fl0;
    const float *zp = has_mul ? s.mul + (size_t)p.n * HW : fl0;
    auto issue = [&](Group &g, uint32_t j0) {
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const uint32_t j = j0 + (uint32_t)i, ri = wid + j * (uint32_t)(TT / 64);
            bool on = j < my_n;
            const uint32_t d = (Cfg::NDIR > 1 && on && ri >= p.len0) ? 1u : 0u;
            const uint32_t rj = on ? ri - (d ? p.len0 : 0u) : 0u;
            int sy, stx;
            uint32_t base = 0;
            if (d ? p.ovf1 : p.ovf0) {
                sy = (int)(rj / (uint32_t)s.tiles_x);
                stx = (int)(rj - (uint32_t)sy * (uint32_t)s.tiles_x);
            } else {
                const uint32_t q = on ? (d ? p.n0 : 0u) + rj : 0u;
                const uint32_t syw = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl_sy[q]);
                sy = (int)(syw & 0xffffffu);
                on = on && ((syw >> 24) & range_mask) != 0u;           // (a piece only loads the segments that touch its column octants)
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rl_w1[q]);
                stx = (int)(w1 >> ROWW_STX);
                if (MODE == 0) base = ((w1 >> ROWW_BASE) & 0x7ffu) + (d ? p.cnt0 : 0u);
            }
            g.sy[i] = on ? sy : -1;
            g.stx[i] = stx;
            g.b0[i] = base;
            g.d[i] = d;
            const float *fl = d ? fl1 : fl0;
            const int sx = stx * TILE_W + lane;
            const bool in = on & (sx < s.W);
            const uint32_t q = in ? (uint32_t)(sy * s.W + sx) : 0u;
            g.fx[i] = fl[q];
            g.fy[i] = fl[HW + q];
            g.z[i] = has_mul ? zp[q] : 0.0f;
            if (G2) { g.l2[i] = s.mul2[(size_t)p.n * HW + q]; g.v2[i] = s.in2[(size_t)p.n * HW + q]; }
        }
    };
    uint32_t wcount = 0;
    auto process = [&](const Group &g) {
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int sy = g.sy[i], sx = g.stx[i] * TILE_W + lane;
            const bool in = (sy >= 0) & (sx < s.W);
            const float X = (float)sx + g.fx[i], Y = (float)sy + g.fy[i];
            const Corners c = corners_at(X, Y);
            const int lx = c.x0 - p.tx0, ly = c.y0 - p.ty0;
            const bool xa = (lx >= p.pca) & (lx < p.pcb) & (c.x0 < s.W), xb = (lx + 1 >= p.pca) & (lx + 1 < p.pcb) & (c.x0 + 1 < s.W);
            const bool ya = (ly >= 0) & (ly < TILE_H) & (c.y0 < s.H), yb = (ly + 1 >= 0) & (ly + 1 < TILE_H) & (c.y0 + 1 < s.H);
            const bool hit = in & c.ok & (xa | xb) & (ya | yb);
            const unsigned long long hm = __ballot(hit);
            const uint32_t pc = (uint32_t)__popcll(hm);
            if (sy < 0) continue;                                      // (wave-uniform: no row segment here)
            uint32_t b0;
            if (MODE == 0) b0 = g.b0[i];
            else if (MODE == 1) {
                b0 = 0;
                if (pc) { if (lane == 0) b0 = atomicAdd(&L.misc[0], pc); b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0); }
            } else b0 = wave_base + wcount;
            wcount += pc;
            if (EMIT) {
                const uint32_t slot = b0 + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
                if (hit && slot >= lo && slot < hi) {
                    const float4 en = make_float4(__uint_as_float((uint32_t)(sy * s.W + sx) | (g.d[i] << 31)), X, Y, g.z[i]);
                    if constexpr (GLB) gent[slot - lo] = en; else L.ent4[slot - lo] = en;
                    if (G2) L.ent2[slot - lo] = make_float2(g.l2[i], g.v2[i]);
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
}

// The count pass of a piece that is walked in several passes: hits per wave -> this wave's first ordinal; returns the piece's total.
// (L.misc[1 .. 8]; contains barriers.)
template <class Cfg>
__device__ __forceinline__ uint32_t wave_bases(const TileLds<Cfg> &L, int tid, uint32_t wave_hits, uint32_t &wave_base) {
    if ((tid & 63) == 0) L.misc[1 + (tid >> 6)] = wave_hits;
    __syncthreads();
    uint32_t all = 0, wb = 0;
#pragma unroll
    for (int w = 0; w < TT / 64; ++w) { const uint32_t c = L.misc[1 + w]; all += c; wb += w < (tid >> 6) ? c : 0u; }
    wave_base = wb;
    __syncthreads();
    return all;
}

template <class Cfg, bool SORTED>
__device__ __forceinline__ void rows_setup(const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid) {
    if constexpr (SORTED) rows_setup_sorted<Cfg>(f, L, p, tid); else rows_setup_unsorted<Cfg>(f, L, p, tid);
}

// What a tile kernel reads once at its start (scalars; see build_records).
struct TileScalars {
    uint32_t hw4;                  // bytes of one plane
    float shift, sc0, sc1;         // subtracted before exp; the two directions' scales
};
__device__ __forceinline__ TileScalars tile_scalars(const TileShared &s, const TileFrame &f) {
    TileScalars k;
    k.hw4 = (uint32_t)(s.H * s.W) * 4u;
    k.shift = (s.mulmode == MUL_EXP_SHIFT) ? s.mulmax[0] : 0.0f;      // (a dependent scalar load: issued first, needed in phase 1a)
    k.sc0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f.scale[0])));
    k.sc1 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f.scale[1])));
    return k;
}
__device__ __forceinline__ rsrc_t sample_planes(const TileShared &s, const Piece &p, uint32_t hw4) {
    return make_rsrc(s.in + (size_t)p.n * s.Cs * ((size_t)s.H * s.W), (uint32_t)s.C * hw4);
}

// A piece in ONE pass (the main kernels: no loop over work): lists -> entries -> records -> planes [cb, ce).  Returns false, with
// nothing written, when the piece turns out to hold more than SEG entries (the pass-by-pass launch takes it).
template <class Cfg, bool SORTED, bool NORM, bool MAXOP, bool G2>
__device__ __forceinline__ bool rows_piece_once(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid,
                                                const TileScalars &k, int cb, int ce) {
    T_STAMP(s, 0);
    L.cnt[tid] = 0;
    if (tid == 0) L.misc[0] = 0;
    rows_setup<Cfg, SORTED>(f, L, p, tid);
    T_STAMP(s, 1);
    uint32_t total;
    if (p.whole && !p.ovf0 && !p.ovf1 && p.cnt0 + p.cnt1 <= (uint32_t)Cfg::SEG) {
        rows_walk<Cfg, 0, true, G2>(s, f, L, p, tid, 0u, 0u, (uint32_t)Cfg::SEG);
        total = p.cnt0 + p.cnt1;                           // exact (the binning kernel), <= SEG (the plan)
        __syncthreads();
    } else {
        rows_walk<Cfg, 1, true, G2>(s, f, L, p, tid, 0u, 0u, (uint32_t)Cfg::SEG);
        __syncthreads();
        total = L.misc[0];
        if (total > (uint32_t)Cfg::SEG) return false;      // (uniform)
    }
    T_STAMP(s, 2);
    T_NOTE(s, 60, total); T_NOTE(s, 61, p.len0 + p.len1); T_NOTE(s, 62, p.pcb - p.pca);
    const rsrc_t rin = sample_planes(s, p, k.hw4);
    EntryRegs<Cfg> e;
    float preA[Cfg::EPT][4], preB[Cfg::EPT][4];
    build_records<Cfg, NORM || (Cfg::NDIR > 1), G2>(s, L, p, tid, total, rin, k.hw4, cb, ce - 1, k.shift, k.sc0, k.sc1, e, preA, preB);
    T_STAMP(s, 6);
    PixelSums sums = {0.0f, 0.0f, 0.0f};
    stream_planes<Cfg, NORM, MAXOP, G2, false>(s, f, L, p, tid, rin, k.hw4, cb, ce, e, preA, preB, sums, true, true);
    T_STAMP(s, 59);
    return true;
}

// A piece pass by pass (more than SEG entries: an octant that is a sink by itself; any pathological flow): a count pass gives every
// wave its first ordinal, pass si stages the entries with ordinals [si * SEG, (si + 1) * SEG); every work-item accumulates its output
// pixel through its own earlier stores and normalises in the last pass.
template <class Cfg, bool SORTED, bool NORM, bool MAXOP, bool G2>
__device__ __forceinline__ void rows_piece_passes(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid,
                                                  const TileScalars &k, int cb, int ce) {
    rows_setup<Cfg, SORTED>(f, L, p, tid);
    uint32_t wb;
    const uint32_t all = wave_bases<Cfg>(L, tid, rows_walk<Cfg, 2, false, G2>(s, f, L, p, tid, 0u, 0u, 0u), wb);
    const uint32_t npass = max(1u, (all + (uint32_t)Cfg::SEG - 1u) / (uint32_t)Cfg::SEG);
    const rsrc_t rin = sample_planes(s, p, k.hw4);
    PixelSums sums = {0.0f, 0.0f, 0.0f};
    for (uint32_t si = 0; si < npass; ++si) {
        __syncthreads();
        if (si > 0) rows_setup<Cfg, SORTED>(f, L, p, tid);     // (the lists share LDS with the previous pass's records)
        L.cnt[tid] = 0;
        const uint32_t lo = si * (uint32_t)Cfg::SEG;
        rows_walk<Cfg, 2, true, G2>(s, f, L, p, tid, wb, lo, lo + (uint32_t)Cfg::SEG);
        __syncthreads();
        EntryRegs<Cfg> e;
        float preA[Cfg::EPT][4], preB[Cfg::EPT][4];
        build_records<Cfg, NORM || (Cfg::NDIR > 1), G2>(s, L, p, tid, min((uint32_t)Cfg::SEG, all - lo), rin, k.hw4, cb, ce - 1, k.shift, k.sc0, k.sc1,
                                                         e, preA, preB);
        stream_planes<Cfg, NORM, MAXOP, G2, true>(s, f, L, p, tid, rin, k.hw4, cb, ce, e, preA, preB, sums, si == 0, si + 1 == npass);
    }
    __syncthreads();
}

// block index of a frame's own grid -> work item: workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Groups of
// SLR_XCD_GROUP consecutive items (= neighbouring tiles / pieces) are placed on the same XCD: a tile's column halo is served by that
// XCD's L2 (-12 % HBM fetch).
__device__ __forceinline__ uint32_t xcd_item(uint32_t bx) {
    constexpr uint32_t G = SLR_XCD_GROUP;
    const uint32_t slot = bx >> 3;
    return ((slot / G) * 8u + (bx & 7u)) * G + slot % G;
}

}  // namespace slr
