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

// Lane `lane` (wave-uniform) of five vectors replaced by five wave-uniform values: one VALU instruction per value where
// `if (lane_id == lane) v = val` costs a compare and exec-masked moves.  (clang of this ROCm has no writelane builtin; a VOP3 instruction
// of gfx9 reads one SGPR, so the lane select goes through M0.)
__device__ __forceinline__ void write_lane5(int lane, int &v0, int a0, int &v1, int a1, int &v2, int a2, int &v3, int a3, int &v4, int a4) {
    asm("s_mov_b32 m0, %10\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\tv_writelane_b32 %2, %7, m0\n\t"
        "v_writelane_b32 %3, %8, m0\n\tv_writelane_b32 %4, %9, m0"
        : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4) : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(lane) : "m0");
}

// The same footprint for the binning kernels (rowbin_kernel, rowbin_clip_kernel), which are bound by their instruction count: the
// <= 4 tiles of a source pixel as linear tile indices (-1: none) and the column octants (8 output columns) it touches in the left /
// right tile column.  Same result as footprint_tiles + the octant rules written out in round 4, in half the instructions: range tests
// as unsigned compares, tile coordinates as shifts (used only where the coordinate is >= 0).
struct BinFoot { int t0, t1, t2, t3; uint32_t cm_a, cm_b; };
__device__ __forceinline__ BinFoot bin_footprint(float fx, float fy, int x, int y, int H, int W, int tiles_x) {
    static_assert(TILE_W == 64 && (TILE_H & (TILE_H - 1)) == 0, "shifts below");
    constexpr int SH_H = TILE_H == 8 ? 3 : TILE_H == 4 ? 2 : TILE_H == 16 ? 4 : TILE_H == 2 ? 1 : 0;
    static_assert((1 << SH_H) == TILE_H, "tile height");
    constexpr int NONE = -0x40000000;                                    // "no tile column / row": sums of two of them stay negative
    const float X = (float)x + fx, Y = (float)y + fy;                    // softsplat.py:169-170
    const bool ok = (fabsf(X) < 1073741824.0f) & (fabsf(Y) < 1073741824.0f);
    const uint32_t ux0 = (uint32_t)(ok ? (int)floorf(X) : NONE), uy0 = (uint32_t)(ok ? (int)floorf(Y) : NONE);
    const uint32_t ux1 = ux0 + 1u, uy1 = uy0 + 1u;
    // tile column of corner column x0 / x0 + 1 (NONE outside the image); the second one only when it is another column -- validity lives in
    // the sign of the values, not in booleans (which this compiler materialises in registers and recombines: a third of the function)
    const int ca = ux0 < (uint32_t)W ? (int)(ux0 >> 6) : NONE;
    const int cb0 = ux1 < (uint32_t)W ? (int)(ux1 >> 6) : NONE;
    const bool same_x = cb0 == ca;                                      // (both NONE: nothing lands anyway)
    const int cb = same_x ? NONE : cb0;
    const int ra = uy0 < (uint32_t)H ? (int)(uy0 >> SH_H) * tiles_x : NONE;
    const int rb0 = uy1 < (uint32_t)H ? (int)(uy1 >> SH_H) * tiles_x : NONE;
    const int rb = rb0 == ra ? NONE : rb0;
    BinFoot f;
    f.t0 = max(ra + ca, -1); f.t1 = max(ra + cb, -1); f.t2 = max(rb + ca, -1); f.t3 = max(rb + cb, -1);
    const uint32_t b0 = 1u << ((ux0 >> 3) & 7u), b1 = 1u << ((ux1 >> 3) & 7u);
    f.cm_a = ca >= 0 ? (same_x ? b0 | b1 : b0) : 0u;                   // x0 + 1 in the same tile column: its octant joins column a's mask
    f.cm_b = cb >= 0 ? b1 : 0u;
    return f;
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
