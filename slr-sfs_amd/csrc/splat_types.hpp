// splat_types.hpp -- records shared by the splat kernels of libslrsplat (splat.hip: the one-flow operator; splat_clip.hip: the
// fused two-flow kernel of a clip).
#pragma once
#include "slr_common.hpp"

namespace slr {

// Tiles touched by the footprint of one source pixel: <= 2 tile columns x <= 2 tile rows.
// (scalars, not arrays: dynamically indexed private arrays would be demoted to LDS/scratch)
struct TileSet {
    int txa, txb, tya, tyb;     // candidate tile columns / rows
    bool vxa, vxb, vya, vyb;    // candidate valid (b only when distinct from a)
};

__device__ __forceinline__ TileSet footprint_tiles(const Corners &c, int H, int W) {
    TileSet s;
    const bool xa = c.ok & (c.x0 >= 0) & (c.x0 < W), xb = c.ok & (c.x0 + 1 >= 0) & (c.x0 + 1 < W);
    const bool ya = c.ok & (c.y0 >= 0) & (c.y0 < H), yb = c.ok & (c.y0 + 1 >= 0) & (c.y0 + 1 < H);
    s.txa = c.x0 / TILE_W; s.txb = (c.x0 + 1) / TILE_W;      // only used when in range (>= 0)
    s.tya = c.y0 / TILE_H; s.tyb = (c.y0 + 1) / TILE_H;
    s.vxa = xa; s.vxb = xb & !(xa & (s.txb == s.txa));
    s.vya = ya; s.vyb = yb & !(ya & (s.tyb == s.tya));
    return s;
}

// Everything a tile workgroup needs to know about its work item, in ONE 32-byte record (one scalar
// load instead of a chain of dependent lookups through items -> count/listoff/nseg/partoff).
struct ItemDesc {
    uint32_t tile, seg;          // tile index (n*tiles + tile), segment of its concatenated bin
    uint32_t cnt0, cnt1;         // entries in the bin of flow 0 / flow 1
    uint32_t off0, off1;         // where those bins start in list[0] / list[1]
    uint32_t nseg, partoff;      // segments of the tile (0: this item covers the whole tile, segment by
                                 // segment); first partial slot (multi-segment tiles)
};

// Rows front end: a row segment = 64 consecutive source pixels of one image row (one wave's coalesced load).  A tile's list holds
// up to ROW_CAP of them: image row | column octants of the tile it touches << 24, (x / 64) << 8 | its hits in the tile (<= 64).
constexpr int ROW_CAP = SLR_ROW_CAP;
struct RowRec { uint32_t sy, sx_cnt; };

}  // namespace slr
