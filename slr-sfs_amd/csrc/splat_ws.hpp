// splat_ws.hpp -- layout of the per-flow workspace of the one-flow operator (slr_splat_workspace_bytes) and of slr_synth_group.
#pragma once
#include "slr_common.hpp"
#include "splat_types.hpp"

namespace slr {

struct OpLayout {
    int tiles_x, tiles_y, tiles;          // per sample
    uint32_t nt;                          // N * tiles
    uint32_t items_cap;                   // one-flow work items: up to 4 pieces per tile on average (rows_plan fills items[] from both ends)
    uint32_t items2_cap;                  // two-flow work items of slr_synth_group: up to 8 pieces per tile
    size_t off_rowcnt;    // uint64[nt][2]  (entries << 32) | row segments appended; entries per column octant / 16, 8 bits each
    size_t off_rowinfo;   // uint64[nt][2]  what the rowcnt words held when the plan was made (the plan leaves rowcnt zero for the next binning)
    size_t off_rowlist;   // RowRec[nt][ROW_CAP]  the tile's row segments as rowbin_kernel appended them
    size_t off_items;     // ItemDesc[items_cap]
    size_t off_totals;    // uint32[16]  [0] items, [4] deferred pieces, [5] heavy items, [6] arrivals of the deferred launch; [8..15]: the same for slr_synth_group's plan
    size_t off_defer;     // uint32[items_cap]  pieces of more than a segment (appended by their workgroups)
    size_t off_box;       // SrcBox[nt]  scan front end: destination box of every source tile
    size_t off_ctl;       // uint32[64]  arrival counter (second level) of rowbin_kernel
    size_t off_arrive;    // uint32[ceil(nt / 64)][32]  first-level arrival counters, one per 128-byte line
    size_t off_sink_cnt;  // uint32[items_cap][16]  scan front end, per deferred piece: [0..7] arrivals (| slabs written << 8) of the sink launch's workgroups per channel group, [8] its candidate-pair tasks, [9] its entries, [10] where they start in sink_ent (0xffffffff: not written out), [11] its first slab of the pool (0xffffffff: none)
    size_t off_sink_pool; // float[sink_pool_bytes / 4]  scan front end: slabs of the sink launch (partial sums of a deferred piece per task slot)
    size_t sink_pool_bytes;
    size_t off_sink_ent;  // float4[sink_ent_cap]  scan front end: the entries of the deferred pieces, written out by their workgroups
    uint32_t sink_ent_cap;
    size_t off_rowlist2;  // RowRec[2][nt][ROW_CAP]  slr_synth_group: sorted copies of this and the other workspace's lists
    size_t off_items2;    // ItemDesc[items2_cap]  slr_synth_group's two-flow plan
    size_t off_defer2;    // uint32[items2_cap]
    size_t total;
};

inline OpLayout op_layout(int N, int H, int W) {
    OpLayout L;
    L.tiles_x = (W + TILE_W - 1) / TILE_W;
    L.tiles_y = (H + TILE_H - 1) / TILE_H;
    L.tiles = L.tiles_x * L.tiles_y;
    L.nt = (uint32_t)N * L.tiles;
    L.items_cap = 4u * L.nt < 64u ? 64u : 4u * L.nt;
    L.items2_cap = 8u * L.nt;
    size_t o = 0;
    L.off_rowcnt = o;   o += al256((size_t)L.nt * 16);
    L.off_rowinfo = o;  o += al256((size_t)L.nt * 16);
    L.off_totals = o;   o += 256;
    L.off_ctl = o;      o += 256;
    L.off_arrive = o;   o += al256((size_t)((L.nt + 63) / 64) * 128);
    L.off_box = o;      o += al256((size_t)L.nt * 16);
    L.off_items = o;    o += al256((size_t)L.items_cap * sizeof(ItemDesc));
    L.off_defer = o;    o += al256((size_t)L.items_cap * 4);
    L.off_items2 = o;   o += al256((size_t)L.items2_cap * sizeof(ItemDesc));
    L.off_defer2 = o;   o += al256((size_t)L.items2_cap * 4);
    L.off_sink_cnt = o; o += al256((size_t)L.items_cap * 16 * 4);
    L.off_rowlist = o;  o += al256((size_t)L.nt * ROW_CAP * sizeof(RowRec));
    L.off_rowlist2 = o; o += al256((size_t)2 * L.nt * ROW_CAP * sizeof(RowRec));
    // (only grids the scan front end takes by default -- up to SLR_SCAN_MAX_TILES tiles -- get the full pool; a larger grid forced onto the scan
    //  front end renders its sinks out of the emergency slabs: one workgroup per piece and channel group, as rounds 1-5 did)
    L.sink_pool_bytes = (size_t)(L.nt <= (uint32_t)SLR_SCAN_MAX_TILES ? SLR_SINK_POOL_MB : 8) << 20;
    L.off_sink_pool = o; o += al256(L.sink_pool_bytes);
    L.sink_ent_cap = (uint32_t)(((size_t)(L.nt <= (uint32_t)SLR_SCAN_MAX_TILES ? SLR_SINK_ENT_MB : 1) << 20) / 16);
    L.off_sink_ent = o; o += al256((size_t)L.sink_ent_cap * 16);
    L.total = o;
    return L;
}

struct OpWs {
    OpLayout L;
    char *base;
    unsigned long long *rowcnt, *rowinfo;
    RowRec *rowlist, *rowlist2;
    ItemDesc *items, *items2;
    uint32_t *totals, *defer, *defer2, *ctl, *arrive, *sink_cnt;
    float *sink_pool;
    float4 *sink_ent;
    void *box;
};

int op_ws_open(OpWs &w, int N, int H, int W, void *ws, size_t bytes, const char *who);
int op_check_dims(int N, int C, int H, int W, const char *who);
int plane_group(int C, int H, int W);

}  // namespace slr
