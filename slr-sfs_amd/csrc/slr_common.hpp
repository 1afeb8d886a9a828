// slr_common.hpp -- shared host/device helpers for libslrsplat (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/slr_splat.h"
#include "slr_tuning.hpp"

#define SLR_EXPORT extern "C" __attribute__((visibility("default")))

namespace slr {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define SLR_CHECK_ARG(cond, what)                                              \
    do { if (!(cond)) { slr::set_error("%s: bad argument: %s", __func__, what); return SLR_E_BADARG; } } while (0)

#define SLR_CHECK_HIP(expr)                                                    \
    do { hipError_t e_ = (expr);                                               \
         if (e_ != hipSuccess) { slr::set_error("%s: %s", __func__, hipGetErrorString(e_)); return (int)e_; } } while (0)

#define SLR_CHECK_LAUNCH() SLR_CHECK_HIP(hipGetLastError())

// ---- geometry of the output tiling ------------------------------------------------------
// One workgroup owns a TILE_H x TILE_W block of OUTPUT pixels of one sample ("tile"); its bin
// lists the source pixels whose bilinear footprint touches the tile.  Long bins are cut into segments of SEG entries so that no workgroup gets more than
// ~2x the average work; a tile with several segments is finished by the combine kernel.
constexpr int TILE_W   = 64;      // one wavefront of consecutive x
constexpr int TILE_H   = SLR_TILE_H;
constexpr int TILE_PIX = TILE_W * TILE_H;
constexpr int SEG_ONE  = SLR_EPT_ONE * TILE_PIX;   // segment length, one flow per tile
constexpr int SEG_TWO  = SLR_EPT_TWO * TILE_PIX;   // segment length, forward+backward flows per tile

// Workspace layout (all offsets 256-byte aligned).  `hdr` is zeroed at the start of binning.
struct WsLayout {
    int tiles_x, tiles_y, tiles;          // per sample
    uint32_t nt;                          // N * tiles
    uint32_t part_slots;                  // partial-tile slots available to the plan
    uint32_t items_cap;                   // upper bound of work items (tiles + part_slots)
    uint32_t rows_items_cap;              // ... of the rows front end (pieces by output rows; items[] holds this many)
    size_t off_count;     // uint32[nt]   entries per tile                 (bin)
    size_t off_cursor;    // uint32[nt]   fill cursors                     (bin)
    size_t off_listoff;   // uint32[nt+1] exclusive prefix of count        (bin)
    size_t off_list;      // uint32[4*N*H*W] source pixel indices          (bin)
    size_t off_nseg;      // uint32[nt]   segments per tile                (plan)
    size_t off_partoff;   // uint32[nt]   first partial slot of the tile   (plan)
    size_t off_multi;     // uint32[nt]   compact list of multi-segment tiles (plan)
    size_t off_whole;     // uint32[4*nt] work items that cover a whole over-budget tile (plan); rows front end: pieces of more than a segment
    size_t off_items;     // ItemDesc[items_cap] (32 B per work item)           (plan)
    size_t off_totals;    // uint32[4]    total items, total partial slots (plan)
    size_t off_box;       // SrcBox[nt]   destination box of every source tile (scan front end: no bins, no plan)
    size_t off_ctl;       // uint32[4]    scan front end, segment sharing: queue head, tail, partial slots used
    size_t off_queue;     // uint64[part_slots] work words (tile, segment, segments, first slot, channel group)
    size_t off_arrive;    // uint32[4 * part_slots] arrivals per shared tile (rows front end: per channel group)
    size_t off_rowcnt;    // uint64[nt][2] rows front end: (entries << 32) | row segments appended; entries per column octant / 16, 8 bits each
    size_t off_rowlist;   // uint32[nt][SLR_ROW_CAP][2]  (row segment, its entries in the tile)
    size_t off_trash;     // float[planes][TILE_PIX]  sink for work-items outside the image
    size_t off_partial;   // float[part_slots][planes][TILE_PIX]           (main -> combine)
    size_t part_stride;   // floats per partial slot = planes * TILE_PIX
    size_t total;
};

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// C = number of value planes that will be splatted with this workspace (0 = bins only).
inline WsLayout ws_layout(int N, int C, int H, int W) {
    WsLayout L;
    L.tiles_x = (W + TILE_W - 1) / TILE_W;
    L.tiles_y = (H + TILE_H - 1) / TILE_H;
    L.tiles = L.tiles_x * L.tiles_y;
    L.nt = (uint32_t)N * L.tiles;
    L.part_slots = L.nt < 64 ? 64 : L.nt;       // one partial-tile slot per tile on average
    L.items_cap = L.nt + L.part_slots;
    L.rows_items_cap = L.items_cap > 4u * L.nt ? L.items_cap : 4u * L.nt;      // rows front end: up to 8 pieces per tile
    size_t o = 0;
    L.off_count = o;   o += al256((size_t)L.nt * 4);
    L.off_cursor = o;  o += al256((size_t)L.nt * 4);
    L.off_listoff = o; o += al256(((size_t)L.nt + 1) * 4);
    L.off_list = o;    o += al256((size_t)4 * N * H * W * 4);
    L.off_nseg = o;    o += al256((size_t)L.nt * 4);
    L.off_partoff = o; o += al256((size_t)L.nt * 4);
    L.off_multi = o;   o += al256((size_t)L.nt * 4);
    L.off_whole = o;   o += al256((size_t)L.nt * 4 * 4);      // (rows front end: up to 4 * nt pieces may ask for the pass-by-pass launch)
    L.off_items = o;   o += al256((size_t)L.rows_items_cap * 32);
    L.off_totals = o;  o += 256;
    L.off_box = o;     o += al256((size_t)L.nt * 16);
    L.off_ctl = o;     o += 256;
    L.off_queue = o;   o += al256((size_t)L.part_slots * 8);
    L.off_arrive = o;  o += al256((size_t)L.part_slots * 4 * 4);
    L.off_rowcnt = o;  o += al256((size_t)L.nt * 16);
    L.off_rowlist = o; o += al256((size_t)L.nt * SLR_ROW_CAP * 8);
    // C value planes (rounded up to whole chunks of 4: the scan front end stores a chunk per 16-byte word) + the normaliser plane
    L.part_stride = (size_t)((C + 3) / 4 * 4 + 1) * TILE_PIX;
    L.off_trash = o;   o += al256(L.part_stride * 4);
    L.off_partial = o;
    o += al256((size_t)L.part_slots * L.part_stride * 4);
    L.total = o;
    return L;
}

// ---- bilinear footprint ------------------------------------------------------------------
// Restates models/softsplat.py:169-184 (target coordinate, NW corner, 4 weights).
struct Corners {
    int x0, y0;
    float w[4];        // NW, NE, SW, SE
    bool ok;           // coordinate representable (finite, |.| < 2^30)
};

// (X, Y) = the target coordinate (float)x + fx, (float)y + fy of softsplat.py:169-170
__device__ __forceinline__ Corners corners_at(float X, float Y) {
    Corners c;
    c.ok = (fabsf(X) < 1073741824.0f) && (fabsf(Y) < 1073741824.0f);
    float flx = floorf(X), fly = floorf(Y);
    c.x0 = c.ok ? (int)flx : 0;
    c.y0 = c.ok ? (int)fly : 0;
    float x1 = (float)(c.x0 + 1), y1 = (float)(c.y0 + 1), x0f = (float)c.x0, y0f = (float)c.y0;
    c.w[0] = (x1 - X) * (y1 - Y);
    c.w[1] = (X - x0f) * (y1 - Y);
    c.w[2] = (x1 - X) * (Y - y0f);
    c.w[3] = (X - x0f) * (Y - y0f);
    return c;
}

__device__ __forceinline__ Corners make_corners(float fx, float fy, int x, int y) {
    return corners_at((float)x + fx, (float)y + fy);
}

__device__ __forceinline__ bool in_image(int cx, int cy, int H, int W) {
    return (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H);
}

}  // namespace slr
