// splat_tile.hpp -- what every tile kernel does once the ENTRIES of its piece of work sit in LDS (gfx950):
//   build_records   phase 1: per-OUTPUT-pixel lists of (entry, weight) records, built with integer LDS atomics (once per entry,
//                   not per channel) -- the scatter turned into a gather inside the tile;
//   stream_planes   phase 2: the chunk pipeline -- the entries' source values staged in LDS four planes at a time (plane loads
//                   two chunks ahead, issued between the steps of the gather), every work-item accumulates its own output pixel in
//                   registers, normalises with ONE reciprocal, stores: every output byte written once, never read, never zeroed.
// A piece of work = an 8x64 OUTPUT tile, or a range of its columns (heavy tiles are cut by output columns: a piece owns its pixels,
// nothing is summed across workgroups), of one sample.  An entry = a source pixel whose 2x2 bilinear footprint touches the piece:
// (pixel | direction << 31, target X, target Y, weight logit) in the entry array `ent4` (+ the second weight group's logit and value).
// Where the entries come from is the front end's business: row-segment lists (splat_rows.hpp; the clip kernel and the one-flow
// operator on large grids) or a scan of the flow itself (splat_op.hip, small grids).
// Replaces models/softsplat.py:157-202 (kernel_Softsplat_updateOutput), :390-424, :665-690 and the model-side weighting /
// two-direction accumulation / normalisation of forward_flow (animating_softmax_splating.py:849-924).
#pragma once
#include "splat_core.hpp"

#include <type_traits>

namespace slr {

constexpr int TT = TILE_PIX;                       // work-items per workgroup = output pixels of a tile

typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 WRec;                                   // a record in registers: .x weight, .y the staged entry's byte offset (bits)

enum { MUL_ONE = 0, MUL_PLANE = 1, MUL_EXP = 2, MUL_EXP_SHIFT = 3 };

// ---- shape of a kernel family -------------------------------------------------------------------------------------------------
// NDIR flows per tile (the fused clip kernel: forward + backward displacement map), EPT entries per work-item (a workgroup stages
// SEG = EPT * 512 entries at once), records as (u16 entry, f32 weight) in two arrays (REC6: 6 bytes) or as one 8-byte word, KREG
// records of an output pixel kept in registers across the chunks.  LDS: counts | scan words | misc | offsets | records | staged
// values (+ the all-zero slot).  One flow: EPT 2, REC6 -> 46 KiB, three workgroups per CU; two flows: EPT 3, 8-byte records
// (their lists are twice as long: one ds_read_b64 per record beats two reads) -> 79 KiB, two per CU.
// B4: the value planes are plane-blocked by 4 in memory ([C/4][H][W][4], slr_pack_planes4: the clip path packs its feature planes once per
// clip): a chunk's 4 planes of an entry are ONE 16-byte load that arrives as the staged float4 -- a quarter of the load instructions.
// BAL: 17 KiB more of LDS (a float4 per work-item, a float4 per output pixel, a byte per work-item) for the BALANCED gather of tiles whose
// records pile onto a few pixels (BalLane below; the scan front end's kernels).
template <int NDIR_, int EPT_, bool REC6_, int KREG_, bool B4_ = false, bool BAL_ = false>
struct TileCfg {
    static constexpr int NDIR = NDIR_, EPT = EPT_, KREG = KREG_, CHUNK = 4;
    static constexpr bool REC6 = REC6_, B4 = B4_, BAL = BAL_;
    static_assert(!B4_ || EPT_ <= 4, "one 16-byte load per gather stop");
    static constexpr int SEG = EPT * TT;
    static constexpr int RECCAP = 4 * SEG + TT;    // <= 4 records per entry + one pad per pixel (odd list lengths: bank spreading)
    static constexpr uint32_t NULL_E = SEG;        // staged-entry index of the all-zero slot
    static_assert(!REC6_ || (EPT_ * TILE_PIX + 1) * 16 <= 65536, "byte offsets of staged entries travel in 16 bits (6-byte records)");
    static constexpr uint32_t NULLREC = RECCAP - 1;   // a record (all-zero slot, weight 0) that no list owns (the last pixel's pad)
    static constexpr size_t HEAD = (size_t)(TT + 16 + 16 + TT / 2) * 4 + (BAL_ ? (size_t)TT * 32 + (size_t)TT * 2 : 0);
    static constexpr size_t REC_BYTES = ((size_t)RECCAP * (REC6 ? 6 : 8) + 15) & ~(size_t)15;
    static constexpr size_t LDS_BYTES = HEAD + REC_BYTES + (size_t)(SEG + 1) * 16;
    static_assert(HEAD % 16 == 0, "records and staged values are 16-byte aligned");
    static_assert(SEG < 65536, "entry indices travel in 16 bits");
};

// ---- kernel arguments ---------------------------------------------------------------------------------------------------------
struct TileShared {                // the same for every frame of a launch
    const float *in;               // [N,C,H,W] value planes
    const float *mul;              // [N,1,H,W] weight plane (metric / Z), or nullptr
    const float *mulmax;           // device scalar subtracted before exp (MUL_EXP_SHIFT), or nullptr
    const float *in2, *mul2;       // second weight group: one value plane with its own weight logits (G2 instantiations)
    int N, C, H, W, tiles_x, tiles;   // tiles per sample
    int Cs;                        // planes of a sample IN MEMORY (its stride): C, or more when a launch covers a plane group of a larger stack
    int mulmode, mulmode2, norm_mode;
    float eps, init;               // normaliser clamp; start value of the maximum splat
    long long *trace;              // development builds (-DSLR_TRACE): 64 time stamps per workgroup, or nullptr
};
struct SrcBox { int x0, x1, y0, y1; };             // scan front end: inclusive range of a source tile's footprint NW corners; x0 > x1: none
struct TileFrame {
    const float *flow[2];                  // displacement map(s) [N,2,H,W]
    const RowRec *rowlist[2];              // rows front end: [N * tiles][ROW_CAP]
    const SrcBox *box;                     // scan front end: [N * tiles]
    const ItemDesc *items;                 // the work plan
    uint32_t *totals, *defer;              // [>= 8]: [0] items, [4] deferred pieces, [5] arrivals of the deferred launch, [6] heavy items; [items_cap]
    float *out, *out2, *norm_out;          // [N,C,H,W], [N,1,H,W] (G2), [N,1,H,W] or nullptr
    float scale[2];                        // alpha, 1 - alpha (one flow: 1)
    uint32_t grid, items_cap;              // blocks of this frame (multiple of 8 * XCD group); size of items[]
};

#ifdef SLR_TRACE      // development aid: per-workgroup phase time stamps (shader clock) into TileShared.trace (tools/dev/trace_clip.py)
#define T_STAMP(s_, slot) do { if ((s_).trace && threadIdx.x == 0) (s_).trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (slot)] = clock64(); } while (0)
#define T_NOTE(s_, slot, v) do { if ((s_).trace && threadIdx.x == 0) (s_).trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (slot)] = (long long)(v); } while (0)
#else
#define T_STAMP(s_, slot) do { } while (0)
#define T_NOTE(s_, slot, v) do { } while (0)
#endif

// ---- LDS ----------------------------------------------------------------------------------------------------------------------
// Aliases: the row lists (and the second group's entry words) sit in the record area (dead before the records are written), the
// entry array IS the staging area (dead before the first chunk is staged; a work-item's special-chunk slots are its own entry slots).
template <class Cfg>
struct TileLds {
    uint32_t *cnt, *wsum, *misc;   // [TT] records per output pixel; [16] scan words; [16] counters
    uint16_t *off;                 // [TT] exclusive prefix of the (padded) counts
    float4 *bcarry, *bsum;         // BAL: [TT] the open sum a work-item's positions end with; [TT] the sums of 4 planes per output pixel
    uint16_t *bflag;               // ... [TT] per work-item: 1 = a pixel's region starts or ends among its positions
    uint2 *rec8;                   // !REC6: [RECCAP] (entry, weight bits)
    float *rec_w;                  // REC6: [RECCAP] weights ...
    uint16_t *rec_e;               // ... and [RECCAP] entry indices
    float4 *val4;                  // [SEG + 1] staged values of 4 planes per entry
    uint32_t *rl;                  // row-list words (splat_rows.hpp): [4][NDIR * ROW_CAP]
    float4 *ent4;                  // entries (= val4)
    float2 *ent2;                  // G2: (weight logit, value) of the second group per entry

    __device__ __forceinline__ explicit TileLds(uint32_t *smem) {
        cnt = smem;
        wsum = smem + TT;
        misc = smem + TT + 16;
        off = reinterpret_cast<uint16_t *>(smem + TT + 32);
        bcarry = reinterpret_cast<float4 *>(smem + TT + 32 + TT / 2);
        bsum = bcarry + TT;
        bflag = reinterpret_cast<uint16_t *>(bsum + TT);
        char *r = reinterpret_cast<char *>(smem) + Cfg::HEAD;
        rec8 = reinterpret_cast<uint2 *>(r);
        rec_w = reinterpret_cast<float *>(r);
        rec_e = reinterpret_cast<uint16_t *>(rec_w + Cfg::RECCAP);
        val4 = reinterpret_cast<float4 *>(r + Cfg::REC_BYTES);
        rl = reinterpret_cast<uint32_t *>(r);
        ent4 = val4;
        ent2 = reinterpret_cast<float2 *>(rl + 4 * Cfg::NDIR * ROW_CAP);
    }
    // A record = (weight, BYTE offset of the staged entry in val4): the weight comes first -- an 8-byte record read lands in an even
    // register pair with the weight in its LOW register, the one operand form in which v_pk_fma_f32 may broadcast it (see accum4).
    __device__ __forceinline__ void rec_put(uint32_t i, uint32_t e, float w) const {
        if constexpr (Cfg::REC6) { rec_e[i] = (uint16_t)(e * 16u); rec_w[i] = w; } else rec8[i] = make_uint2(__float_as_uint(w), e * 16u);
    }
    __device__ __forceinline__ WRec rec_get(uint32_t i) const {
        WRec r;
        if constexpr (Cfg::REC6) { r.x = rec_w[i]; r.y = __uint_as_float((uint32_t)rec_e[i]); }
        else { const uint2 q = rec8[i]; r.x = __uint_as_float(q.x); r.y = __uint_as_float(q.y); }
        return r;
    }
    __device__ __forceinline__ float rec_weight(uint32_t i) const {
        if constexpr (Cfg::REC6) return rec_w[i]; else return __uint_as_float(rec8[i].x);
    }
    __device__ __forceinline__ float4 staged(const WRec &r) const {       // the 4 staged planes of the record's entry: one ds_read_b128
        return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(val4) + __float_as_uint(r.y));
    }
};

// ---- a piece of work ----------------------------------------------------------------------------------------------------------
struct Piece {
    uint32_t tile;                 // n * tiles + tile of the sample
    int n;                         // sample
    int ty0, tx0;                  // the tile's first output row / column
    int pca, pcb;                  // the piece's output columns [pca, pcb) of the tile (tile-local)
    uint32_t cnt0, cnt1;           // rows front end: exact entries of the TILE per direction
    uint32_t len0, len1;           // ... row segments to walk per direction (the whole sample's if the list overflowed)
    uint32_t n0;                   // ... list entries of direction 0 in LDS (direction 1 follows them)
    bool ovf0, ovf1;               // ... the direction's list overflowed ROW_CAP: every row segment of the sample is scanned
    bool whole;                    // all 8 column octants
};

// What a work-item keeps of its EPT entries for the chunk pipeline.
template <class Cfg>
struct EntryRegs {
    uint32_t off[Cfg::EPT];        // byte offset of the source pixel inside a plane of its sample
    float m[Cfg::EPT];             // G2: the first group's weight of the entry (applied when its values are staged)
};

template <class Cfg>
__device__ __forceinline__ void prefetch_planes(rsrc_t rin, const EntryRegs<Cfg> &e, float (&pre)[Cfg::EPT][4], int c0, int cmax, uint32_t hw4) {
    if constexpr (Cfg::B4) {                           // (C % 4 == 0, chunks start at multiples of 4: past the last chunk re-read it)
        const uint32_t soff = (uint32_t)(min(c0, cmax - 3) >> 2) * (hw4 * 4u);
#pragma unroll
        for (int j = 0; j < Cfg::EPT; ++j) {
            const float4 v = buf_ld4(rin, e.off[j] * 4u, soff);
            pre[j][0] = v.x; pre[j][1] = v.y; pre[j][2] = v.z; pre[j][3] = v.w;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t soff = (uint32_t)min(c0 + u, cmax) * hw4;           // (planes past the last re-read it)
#pragma unroll
        for (int j = 0; j < Cfg::EPT; ++j) pre[j][u] = buf_ld(rin, e.off[j], soff);
    }
}

// Phase 1: the piece's `total` entries (in the entry array) -> per-output-pixel record lists.
//   1a  footprint of this work-item's entries, weight m = 1 | metric | exp(metric - max) (x alpha | 1 - alpha with two flows); one
//       LDS atomic per corner reserves a slot in that output pixel's list -- all four of an entry go out before the first result is
//       looked at, without branches (a corner outside the piece adds 0 to this work-item's own counter); the plane loads of the
//       first two chunks [c0, c0 + 8) are issued as soon as the source pixels are known;
//   1b  workgroup scan of the list lengths (padded to odd: the lanes' list walks then start on different banks);
//   1c  the (entry, weight) records are scattered into the lists.
// WEIGHTED: the record weights carry m (normalised / fused kernels); else m is not applied (plain summation, maximum splat: the
// pure bilinear weights).  G2 (a second weight group shares the records): the records keep the PURE bilinear weights, m multiplies
// the first group's values when they are staged, and the entry's slot of a special chunk carries  m | in2 * m2 | m2.
template <class Cfg, bool WEIGHTED, bool G2>
__device__ __forceinline__ void build_records(const TileShared &s, const TileLds<Cfg> &L, const Piece &p, int tid, uint32_t total,
                                              rsrc_t rin, uint32_t hw4, int c0, int cmax, float shift, float sc0, float sc1,
                                              EntryRegs<Cfg> &e, float (&preA)[Cfg::EPT][4], float (&preB)[Cfg::EPT][4]) {
    constexpr int EPT = Cfg::EPT;
    uint32_t dir[EPT];
    float X[EPT], Y[EPT], mm[EPT], l2[EPT], v2[EPT];
    bool val[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t k = (uint32_t)tid + (uint32_t)j * TT;
        val[j] = k < total;
        float4 en = make_float4(0.f, 0.f, 0.f, 0.f);
        if (val[j]) en = L.ent4[k];
        const uint32_t pw = __float_as_uint(en.x);
        dir[j] = pw >> 31;
        e.off[j] = (pw & 0x7fffffffu) * 4u;
        e.m[j] = 1.0f;
        X[j] = en.y; Y[j] = en.z; mm[j] = en.w;
        if (G2) { float2 e2 = make_float2(0.f, 0.f); if (val[j]) e2 = L.ent2[k]; l2[j] = e2.x; v2[j] = e2.y; }
    }
    prefetch_planes<Cfg>(rin, e, preA, c0, cmax, hw4);
    prefetch_planes<Cfg>(rin, e, preB, c0 + 4, cmax, hw4);
    T_STAMP(s, 3);
    uint32_t ts[EPT][4];                              // (output pixel << 16) | slot, 0xffffffff = corner not in the piece
    float w[EPT][4];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ts[j][k] = 0xffffffffu; w[j][k] = 0.0f; }
        if (G2 && !val[j]) L.val4[tid + j * TT] = make_float4(0.f, 0.f, 0.f, 0.f);      // (no record points here)
        if (!val[j]) continue;
        const Corners c = corners_at(X[j], Y[j]);
        // (two scalars read at the kernel's start: taken from the frame's arguments here, by a per-lane index or select, they
        //  become a vector memory load behind the plane loads)
        const float sc = (Cfg::NDIR > 1 && dir[j]) ? sc1 : sc0;
        float m = sc;
        if (WEIGHTED) {
            if (s.mulmode == MUL_PLANE) m = mm[j] * m;
            else if (s.mulmode >= MUL_EXP) m = expf(mm[j] - shift) * m;
        }
        if (G2) {
            float m2 = sc;
            m2 = s.mulmode2 == MUL_PLANE ? l2[j] * m2 : expf(l2[j]) * m2;
            L.val4[tid + j * TT] = make_float4(m, v2[j] * m2, m2, 0.0f);
            e.m[j] = m;
            m = 1.0f;
        }
        const int lx = c.x0 - p.tx0, ly = c.y0 - p.ty0;
        const bool xa = c.ok & (lx >= p.pca) & (lx < p.pcb) & (c.x0 < s.W);
        const bool xb = c.ok & (lx + 1 >= p.pca) & (lx + 1 < p.pcb) & (c.x0 + 1 < s.W);
        const bool ya = (ly >= 0) & (ly < TILE_H) & (c.y0 < s.H);
        const bool yb = (ly + 1 >= 0) & (ly + 1 < TILE_H) & (c.y0 + 1 < s.H);
        const int oc = ly * TILE_W + lx - p.pca;      // (a piece's columns start at lane 0 of the row's wave)
        const bool kb[4] = {bool(xa & ya), bool(xb & ya), bool(xa & yb), bool(xb & yb)};
        const int tg[4] = {oc, oc + 1, oc + TILE_W, oc + TILE_W + 1};
        uint32_t slot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) slot[k] = atomicAdd(&L.cnt[kb[k] ? tg[k] : tid], kb[k] ? 1u : 0u);      // ds_add_rtn_u32
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ts[j][k] = kb[k] ? ((uint32_t)tg[k] << 16) | slot[k] : 0xffffffffu;
            w[j][k] = kb[k] ? ((WEIGHTED || Cfg::NDIR > 1) ? m * c.w[k] : c.w[k]) : 0.0f;
        }
    }
    T_STAMP(s, 4);
    __syncthreads();
    T_STAMP(s, 5);
    {                                                 // 1b
        const uint32_t v = L.cnt[tid] | 1u;
        const uint32_t ex = block_excl_scan(v, L.wsum, tid);
        L.off[tid] = (uint16_t)ex;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j)                     // 1c
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ts[j][k] != 0xffffffffu) L.rec_put(L.off[ts[j][k] >> 16] + (ts[j][k] & 0xffffu), (uint32_t)tid + (uint32_t)j * TT, w[j][k]);
    if (tid == 0) { L.val4[Cfg::NULL_E] = make_float4(0.f, 0.f, 0.f, 0.f); L.rec_put(Cfg::NULLREC, Cfg::NULL_E, 0.0f); }
    __syncthreads();
}

// The record list of this work-item's output pixel for the gather: the first KREG records live in registers for the whole chunk
// loop (missing ones are the NULL record: all-zero slot, weight 0 -- fma(0, 0, acc) == acc, the gather has no selects); what is left
// of a list far longer than the wave's average (a "sink" pixel) is walked by the whole wave, lane-strided, and wave-reduced.
// A NARROW piece (<= 32 output columns: a cut of a ridge tile, 750 entries piled onto 64 pixels, lists of 150 records) keeps the tile's
// shape -- wave w = output row w -- and gives every output pixel G = 2, 4 or 8 lanes (32, 16, 8 columns): lane g of a pixel takes
// records g, g + G, ... of its list and the G partial sums are added up through log2(G) cross-lane steps.  One lane per pixel
// walked such lists for 170-250 us while the rest of its wave sat idle; with G lanes the longest workgroup of Euler t=59 takes 81.
template <class Cfg>
struct PixelList {
    uint32_t r0, rl, r1;           // own records r0, r0 + G, ... < rl; cooperative rest [rl, r1)
    unsigned long long heavy;      // lanes of this wave whose rest the wave walks together
    int g_log;                     // log2 of the lanes per output pixel
    int pid;                       // the output pixel (index in the piece's 8 x 64 frame) this work-item serves
    WRec rc[Cfg::KREG];            // the first KREG records of the list
};

__device__ __forceinline__ int lane_group_log(const Piece &p) {
    const int pw = p.pcb - p.pca;
    return !SLR_ROWS_GROUP ? 0 : pw <= 8 ? 3 : pw <= 16 ? 2 : (pw <= 32 && SLR_ROWS_GROUP > 1) ? 1 : 0;
}

template <class Cfg>
__device__ __forceinline__ PixelList<Cfg> pixel_list(const TileLds<Cfg> &L, int tid, int g_log, int heavy_max = SLR_HEAVY_MAX) {
    PixelList<Cfg> g;
    g.g_log = g_log;
    g.pid = (tid & ~63) | ((tid & 63) >> g_log);
    const uint32_t first = L.off[g.pid];
    g.r0 = first + ((uint32_t)tid & ((1u << g_log) - 1u));
    g.r1 = first + L.cnt[g.pid];
    uint32_t wave_recs = g.r1 - first;                 // records of this wave's output pixels (x G with lane groups: they walk their own)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_recs += __shfl_xor(wave_recs, d);
    // own share: twice the wave's average list length (a uniformly compressed region stays per-lane), at least SLR_LMAX; what is left
    // of a longer list goes to the whole wave once it exceeds SLR_HEAVY_SLACK records (a cooperative pass costs ~50 cross-lane
    // operations per chunk, a lane walking alone ~10 per record while the other 63 wait)
    const uint32_t own = max((uint32_t)SLR_LMAX, 2u * ((wave_recs + 63u) >> 6));
    g.rl = (g.r1 - first >= own + (uint32_t)SLR_HEAVY_SLACK) ? first + own : g.r1;
    g.heavy = __ballot(g.r1 > g.rl);
    // (the cooperative passes run one after the other; with many long lists in one wave -- or lane groups -- every lane walks its own)
    if (heavy_max >= 64 && !g_log) {
        // sink tasks (whole tiles whose entries pile onto some pixels, splat_op.hip): how long a list a lane walks alone is chosen per wave --
        // a lane's record costs ~32 cycles of its wave, a cooperative pass ~512 (cross-lane reductions of 4 planes) + its share; of the
        // thresholds 16 .. 256 and "everybody alone" the cheapest by that model.  (The fixed rule -- twice the wave's average, as many
        // cooperative passes as it takes -- made tasks with ~40 long lists in a wave take 12 us per chunk.)
        const uint32_t len = g.r1 - first;
        g.rl = g.r1; g.heavy = 0ull;
        if (__ballot(len >= 16u + (uint32_t)SLR_HEAVY_SLACK)) {            // (uniform; ordinary waves have no such list and pay one ballot)
            uint32_t mx = len;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
            uint32_t best_t = mx, best_c = mx * 32u;
#pragma unroll
            for (uint32_t t = 16; t <= 256; t <<= 1) {
                const uint32_t nh = (uint32_t)__popcll(__ballot(len >= t + (uint32_t)SLR_HEAVY_SLACK));
                const uint32_t c = t * 32u + nh * 512u;
                if (c < best_c) { best_c = c; best_t = t; }
            }
            g.rl = len >= best_t + (uint32_t)SLR_HEAVY_SLACK ? first + best_t : g.r1;
            g.heavy = __ballot(g.r1 > g.rl);
        }
    } else if (__popcll(g.heavy) > heavy_max || g_log) { g.rl = g.r1; g.heavy = 0ull; }
#pragma unroll
    for (int k = 0; k < Cfg::KREG; ++k) {
        const uint32_t r = g.r0 + ((uint32_t)k << g_log);
        g.rc[k] = L.rec_get(r < g.rl ? r : Cfg::NULLREC);
    }
    return g;
}

template <bool MAXOP>
__device__ __forceinline__ float group_reduce(float v, int g_log) {    // the G lanes of a pixel add up (every one of them ends up with the sum)
    if (g_log >= 3) { const float o = __shfl_xor(v, 4); v = MAXOP ? fmaxf(v, o) : v + o; }
    if (g_log >= 2) { const float o = __shfl_xor(v, 2); v = MAXOP ? fmaxf(v, o) : v + o; }
    if (g_log >= 1) { const float o = __shfl_xor(v, 1); v = MAXOP ? fmaxf(v, o) : v + o; }
    return v;
}

// Four planes' accumulators as two register pairs.  acc (+)= staged value * weight; the sum is two v_pk_fma_f32 (two FMAs each, the
// same single rounding as v_fma_f32) with the record's weight broadcast from the LOW register of its pair (op_sel_hi:[1,0,1]).  That is
// the operand form measured correct next to another wave's MFMAs; the high-register broadcast (op_sel:[0,1,0]), which is what the
// compiler's own pairing of scalar FMAs produces for every second weight, is the one that returned wrong low halves (DESIGN.md 3.2,
// tools/ubench/pkfma_repro.hip) -- the library is built without SLP vectorisation, and tests/test_abi_and_host.py disassembles the
// code objects: every packed-fp32 instruction in them must be this form.
struct Acc4 { f2 lo, hi; };

template <bool MAXOP>
__device__ __forceinline__ Acc4 acc4_init(float init) {
    Acc4 a;
    a.lo.x = a.lo.y = a.hi.x = a.hi.y = MAXOP ? init : 0.0f;
    return a;
}

template <bool MAXOP>
__device__ __forceinline__ void accum4(Acc4 &a, const float4 &v, const WRec &r, bool on) {
    if (MAXOP) {
        const float w = r.x;
        a.lo.x = fmaxf(on ? v.x * w : -INFINITY, a.lo.x); a.lo.y = fmaxf(on ? v.y * w : -INFINITY, a.lo.y);
        a.hi.x = fmaxf(on ? v.z * w : -INFINITY, a.hi.x); a.hi.y = fmaxf(on ? v.w * w : -INFINITY, a.hi.y);
    } else {
        const float w = r.x;
        a.lo.x = __builtin_fmaf(v.x, w, a.lo.x); a.lo.y = __builtin_fmaf(v.y, w, a.lo.y);
        a.hi.x = __builtin_fmaf(v.z, w, a.hi.x); a.hi.y = __builtin_fmaf(v.w, w, a.hi.y);
    }
}

// acc[u] (+)= over the pixel's records of staged value[u] * weight, for the 4 planes staged in LDS (MAXOP: maximum).
// between(k), k = 0..3: called at four points of the gather -- the chunk pipeline issues the plane loads of a later chunk there, a
// few at a time: the waves of a workgroup run in step, and a burst of loads per wave waits for the texture addresser (0.28 us per
// chunk, measured) while the LDS pipe idles, then the LDS reads of the gather queue up while the addresser idles.
// COOPN > 1 (scan / sink kernels: lists of ~1000 records on a pixel): the cooperative walks read COOPN records per lane and trip.
template <class Cfg, bool MAXOP, int COOPN = 1, typename F>
__device__ __forceinline__ void gather_chunk(const TileLds<Cfg> &L, const PixelList<Cfg> &g, int lane, float init, float (&acc)[4], F &&between) {
    constexpr int RB = 4, KREG = Cfg::KREG;
    constexpr uint32_t NULL_B = Cfg::NULL_E * 16u;
    Acc4 a = acc4_init<MAXOP>(init);
    {
        between(0);
        float4 v[KREG];
#pragma unroll
        for (int k = 0; k < KREG; ++k) v[k] = L.staged(g.rc[k]);     // ds_read_b128: 4 planes per LDS instruction
        between(1);
#pragma unroll
        for (int k = 0; k < KREG; ++k) accum4<MAXOP>(a, v[k], g.rc[k], __float_as_uint(g.rc[k].y) != NULL_B);
    }
    between(2);
    for (uint32_t r = g.r0 + ((uint32_t)KREG << g.g_log); r < g.rl; r += (uint32_t)RB << g.g_log) {
        WRec q[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) { const uint32_t i = r + ((uint32_t)k << g.g_log); q[k] = L.rec_get(i < g.rl ? i : Cfg::NULLREC); }
        float4 v[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) v[k] = L.staged(q[k]);
#pragma unroll
        for (int k = 0; k < RB; ++k) accum4<MAXOP>(a, v[k], q[k], __float_as_uint(q[k].y) != NULL_B);
    }
    between(3);
    for (unsigned long long hv = g.heavy; hv; hv &= hv - 1) {                     // long lists, cooperatively
        const int src = __ffsll((long long)hv) - 1;
        const uint32_t hb = __shfl(g.rl, src), he = __shfl(g.r1, src);
        Acc4 part = acc4_init<MAXOP>(-INFINITY);
        uint32_t r = hb + (uint32_t)lane;
        if constexpr (COOPN > 1) {
            for (; r + 64u * (COOPN - 1) < he; r += 64u * COOPN) {      // (all of them inside the list)
                WRec q[COOPN];
                float4 v[COOPN];
#pragma unroll
                for (int k = 0; k < COOPN; ++k) q[k] = L.rec_get(r + 64u * (uint32_t)k);
#pragma unroll
                for (int k = 0; k < COOPN; ++k) v[k] = L.staged(q[k]);
#pragma unroll
                for (int k = 0; k < COOPN; ++k) accum4<MAXOP>(part, v[k], q[k], true);
            }
        }
        for (; r < he; r += 64) {
            const WRec q = L.rec_get(r);
            accum4<MAXOP>(part, L.staged(q), q, true);
        }
        float pt[4] = {part.lo.x, part.lo.y, part.hi.x, part.hi.y}, tot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = pt[u];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { const float o = __shfl_xor(t, d); t = MAXOP ? fmaxf(t, o) : t + o; }
            tot[u] = t;
        }
        if (lane == src) {
            a.lo.x = MAXOP ? fmaxf(a.lo.x, tot[0]) : a.lo.x + tot[0]; a.lo.y = MAXOP ? fmaxf(a.lo.y, tot[1]) : a.lo.y + tot[1];
            a.hi.x = MAXOP ? fmaxf(a.hi.x, tot[2]) : a.hi.x + tot[2]; a.hi.y = MAXOP ? fmaxf(a.hi.y, tot[3]) : a.hi.y + tot[3];
        }
    }
    acc[0] = a.lo.x; acc[1] = a.lo.y; acc[2] = a.hi.x; acc[3] = a.hi.y;
    if (g.g_log) {                                     // (uniform)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = group_reduce<MAXOP>(acc[u], g.g_log);
    }
}

// ---- the BALANCED gather ------------------------------------------------------------------------------------------------------------
// Euler-integrated flows pile records onto a few output pixels and rows: a tile of 885 entries whose 3500 records sit in ONE tile row keeps
// that row's wave busy for 5-6 us per chunk (~55 records per lane, two dependent LDS reads each) while seven waves idle -- such tiles made
// the scan tile kernel take 55-73 us on smooth flows against 29 on incoherent ones.  The record lists of a tile are contiguous in pixel
// order, so the work is dealt by POSITION: work-item t takes positions [t * S, (t + 1) * S) of the record area (S = positions / 512 <= 9,
// list pads included).  A pixel whose region lies inside one work-item's positions is summed and stored by it (bsum[pixel]); a region cut
// by position ranges is put together without atomics: every work-item leaves the open sum its range ends with in bcarry[t], and after the
// chunk's second barrier the work-item holding the region's END adds its own head, the carries of the ranges in between and the tail of
// the range the region starts in -- in a fixed order: reproducible -- then a third barrier, and every work-item reads its own pixel's sum.
// Chosen per tile, when its longest list exceeds S by SLR_BAL_SLACK records (ordinary tiles keep the register path).
// (Tried first: LDS float atomics into the pixel's sum -- 152 us for the C2-sized smooth case against 91: a pile-up pixel's ~100 flushes
//  hit one address; without the flushes, timing only, 75.  Then the long lists dealt to the tile's waves as cooperative walks: 96.)
struct BalLane {
    uint32_t q0, p0;               // first position of this work-item; the output pixel of that position
    uint32_t valid, endm;          // bit k: position q0 + k holds a record; ... is the last position of its pixel's region
    uint32_t ua;                   // for a work-item that ends a region begun before its range: the work-item whose range the region starts in
    bool cont_in;                  // the first position continues a region begun before this range
    WRec rc[9];                    // the records of its positions (pads and positions past the end: the all-zero record) -- they do not change from
};                                 // chunk to chunk: a chunk's walk is 9 independent LDS reads of staged values

template <class Cfg>
__device__ __forceinline__ BalLane bal_setup(const TileLds<Cfg> &L, int tid, uint32_t S, uint32_t ntot) {
    static_assert(TT == 512, "nine halvings");
    BalLane b;
    b.q0 = (uint32_t)tid * S;
    uint32_t lo = 0, hi = TT - 1;                        // the last pixel whose region starts at or before q0
#pragma unroll
    for (int it = 0; it < 9; ++it) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        const bool le = (uint32_t)L.off[mid] <= b.q0;
        lo = le ? mid : lo; hi = le ? hi : mid - 1u;
    }
    b.p0 = lo;
    uint32_t p = lo, roff = L.off[p], rcnt = L.cnt[p];
    b.valid = 0u; b.endm = 0u;
    b.cont_in = b.q0 < ntot && roff != b.q0;
    for (uint32_t k = 0; k < S; ++k) {
        const uint32_t q = b.q0 + k;
        if (q >= ntot) break;
        if (q >= roff + (rcnt | 1u)) { ++p; roff = L.off[p]; rcnt = L.cnt[p]; }      // (every region holds a position: one step at most)
        b.valid |= (q < roff + rcnt ? 1u : 0u) << k;
        b.endm |= (q + 1u == roff + (rcnt | 1u) ? 1u : 0u) << k;
    }
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) b.rc[k] = L.rec_get(((b.valid >> k) & 1u) ? b.q0 + k : Cfg::NULLREC);
    L.bflag[tid] = (b.endm != 0u || !b.cont_in) ? 1u : 0u;
    __syncthreads();
    b.ua = (uint32_t)tid;
    if (b.cont_in && b.endm) {                           // walk back to the range the region starts in (pass-through ranges in between)
        uint32_t u = (uint32_t)tid - 1u;                  // (cont_in: tid > 0)
        while (!L.bflag[u]) --u;                          // (range 0 never continues a region: the walk ends)
        b.ua = u;
    }
    return b;
}

// one chunk: complete regions -> bsum, the open tail -> bcarry; returns the head (the sum up to the first region end of a range that
// continues a region begun before it)
template <class Cfg, typename F>
__device__ __forceinline__ Acc4 bal_walk(const TileLds<Cfg> &L, const BalLane &b, int tid, F &&between) {
    Acc4 a = acc4_init<false>(0.0f), head = acc4_init<false>(0.0f);
    uint32_t p = b.p0;
    bool first = b.cont_in;
    float4 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (k < 4) between(k);
        v[k] = L.staged(b.rc[k]);
    }
#pragma unroll
    for (uint32_t k = 0; k < 9; ++k) {
        accum4<false>(a, v[k], b.rc[k], true);           // (the all-zero record adds 0 x 0)
        if ((b.endm >> k) & 1u) {
            if (first) head = a; else L.bsum[p] = make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y);
            first = false;
            a = acc4_init<false>(0.0f);
            ++p;
        }
    }
    L.bcarry[tid] = make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y);
    return head;
}

// after the barrier: the regions cut by position ranges, by the work-item that holds their end
template <class Cfg>
__device__ __forceinline__ void bal_join(const TileLds<Cfg> &L, const BalLane &b, int tid, const Acc4 &head) {
    if (b.cont_in && b.endm) {
        float4 t = make_float4(head.lo.x, head.lo.y, head.hi.x, head.hi.y);
        // (8 carries in flight: a pile-up pixel's region spans ~100 ranges; one LDS round trip per carry was 6 us per chunk)
        for (int u = tid - 1; u >= (int)b.ua; u -= 8) {
            float4 c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = L.bcarry[max(u - i, 0)];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (u - i >= (int)b.ua) { t.x += c[i].x; t.y += c[i].y; t.z += c[i].z; t.w += c[i].w; }
        }
        L.bsum[b.p0] = t;
    }
}

// sum of the pixel's record weights (the normaliser when the weights carry m)
template <class Cfg>
__device__ __forceinline__ float weight_sum(const TileLds<Cfg> &L, const PixelList<Cfg> &g, int lane) {
    float nrm = 0.0f;
    for (uint32_t r = g.r0; r < g.rl; r += 1u << g.g_log) nrm += L.rec_weight(r);
    if (g.g_log) nrm = group_reduce<false>(nrm, g.g_log);
    for (unsigned long long hv = g.heavy; hv; hv &= hv - 1) {
        const int src = __ffsll((long long)hv) - 1;
        const uint32_t hb = __shfl(g.rl, src), he = __shfl(g.r1, src);
        float part = 0.0f;
        for (uint32_t r = hb + (uint32_t)lane; r < he; r += 64) part += L.rec_weight(r);
        part = wave_sum(part);
        if (lane == src) nrm += part;
    }
    return nrm;
}

// What a work-item carries from pass to pass of a piece that needs several (one pass otherwise).
struct PixelSums { float nrm, g2_sum, g2_nrm; };

// Phase 2 for one pass over the planes [cb, ce) (the whole stack, or a channel group's share on small grids): [special chunk] ->
// chunk pipeline.  first / last: the pass is the piece's first / last one (ACCUM kernels: a piece of several passes accumulates
// through its own earlier stores and normalises in the last pass).
// SLAB (the sink launch of the scan front end, splat_op.hip; with ACCUM): this workgroup holds only SOME of the piece's entries, batch after
// batch -- other workgroups the rest -- so it accumulates its un-normalised sums in a slab of its own ([planes of its channel group + a
// normaliser row][8 x 64 pixels of the tile], through its own earlier stores) and the piece's last workgroup adds the slabs up.
// ADAPT: the per-wave choice of how long a list a lane walks alone (pixel_list) and 4 records per lane and trip in the cooperative walks.
template <class Cfg, bool NORM, bool MAXOP, bool G2, bool ACCUM, bool SLAB = false, bool ADAPT = SLAB>
__device__ __forceinline__ void stream_planes(const TileShared &s, const TileFrame &f, const TileLds<Cfg> &L, const Piece &p, int tid,
                                              rsrc_t rin, uint32_t hw4, int cb, int ce, const EntryRegs<Cfg> &e,
                                              float (&preA)[Cfg::EPT][4], float (&preB)[Cfg::EPT][4], PixelSums &sums, bool first, bool last,
                                              float *slab = nullptr) {
    static_assert(!SLAB || (!G2 && ACCUM), "sink tasks: one weight group, accumulated batch after batch");
    constexpr int EPT = Cfg::EPT;
    const int lane = tid & 63;
    // (a sink task: whole tiles whose entries pile onto a few pixels -- every list far above the wave's average is walked by the whole wave;
    //  left to their own lanes, lists of ~1000 records made single tasks take 250 us)
    const PixelList<Cfg> g = pixel_list<Cfg>(L, tid, lane_group_log(p), ADAPT ? 64 : SLR_HEAVY_MAX);
    // balanced gather (BAL configurations, one weight group, sums; not the sink tasks, whose pile-ups sit on single pixels: cooperative walks
    // serve those better): by the tile's longest list against the positions per work-item.  Every wave leaves its longest list in LDS now;
    // the decision is taken behind the first chunk's barrier (a barrier of its own cost ordinary tiles ~1 us per call).
    int bal = 0;                                       // 0 undecided, 1 balanced, 2 register path
    uint32_t bal_s = 0;
    BalLane bl = {};
    constexpr bool CAN_BAL = Cfg::BAL && !MAXOP && !G2 && !SLAB;
    if constexpr (CAN_BAL) {
        if (g.g_log == 0) {                            // (uniform; pieces with lane groups keep their own rule)
            uint32_t mx = g.r1 - (uint32_t)L.off[g.pid];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
            if (lane == 0) L.wsum[8 + (tid >> 6)] = mx;
        } else bal = 2;
    }
    const int ly = g.pid / TILE_W, lx = p.pca + g.pid - ly * TILE_W;
    const int oy = p.ty0 + ly, ox = p.tx0 + lx;
    const bool inside = (oy < s.H) & (ox < s.W) & (lx < p.pcb) & ((tid & ((1 << g.g_log) - 1)) == 0);      // (the first lane of a pixel's group stores)
    const uint32_t opix = (uint32_t)(oy * s.W + ox);
    const uint32_t voff = inside ? (SLAB ? (uint32_t)(ly * TILE_W + lx) : opix) * 4u : BUF_OOB;   // (work-items outside the image / the piece: stores dropped)
    const size_t hw = (size_t)s.H * s.W;
    const uint32_t obytes = SLAB ? (uint32_t)TILE_PIX * 4u : hw4;         // bytes of an output plane; a slab's planes start at cb
    const int cbase = SLAB ? cb : 0;
    const rsrc_t rout = SLAB ? make_rsrc(slab, (uint32_t)(ce - cb + 1) * obytes) : make_rsrc(f.out + (size_t)p.n * s.Cs * hw, (uint32_t)s.C * hw4);
    float inv = 1.0f;
    if (NORM) {
        if (G2) {
            // the special chunk (m | in2 * m2 | m2 per entry, staged in phase 1a): both normalisers and the second group's sum
            float a2[4];
            gather_chunk<Cfg, false, 1>(L, g, lane, 0.0f, a2, [](int) {});
            sums.nrm += a2[0]; sums.g2_sum += a2[1]; sums.g2_nrm += a2[2];
            if (last && inside && cb == 0) f.out2[(size_t)p.n * hw + opix] = sums.g2_sum / norm_divisor(sums.g2_nrm, s.norm_mode, s.eps);
            __syncthreads();                          // val4 is overwritten by the first value chunk
        } else {
            sums.nrm += weight_sum<Cfg>(L, g, lane);
        }
        if constexpr (SLAB) {
            buf_st<BUF_SC1>(rout, voff, (uint32_t)(ce - cb) * obytes, sums.nrm);    // the normaliser so far: the slab's last row
        } else {
            if (last && inside && f.norm_out && cb == 0) f.norm_out[(size_t)p.n * hw + opix] = norm_divisor(sums.nrm, s.norm_mode, s.eps);
            inv = 1.0f / norm_divisor(sums.nrm, s.norm_mode, s.eps);       // ONE division per output pixel
        }
    }
    T_STAMP(s, 7);
    T_NOTE(s, 63, g.r1 - g.r0);
    const int cmax = ce - 1;
    // FULL: all 4 planes exist -- every load and store of the body is unconditional, so the compiler knows how many memory
    // operations are younger than the ones it has to wait for and emits s_waitcnt vmcnt(N) with N > 0 (a conditional store anywhere in
    // the loop makes it drain the whole queue at the top of every chunk: the prefetch distance of two chunks becomes one)
    auto chunk = [&](auto full_tag, float (&pre)[EPT][4], int c0) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            L.val4[tid + j * TT] = G2 ? make_float4(pre[j][0] * e.m[j], pre[j][1] * e.m[j], pre[j][2] * e.m[j], pre[j][3] * e.m[j])
                                      : make_float4(pre[j][0], pre[j][1], pre[j][2], pre[j][3]);
        if (c0 - cb < 32) T_STAMP(s, 8 + 6 * ((c0 - cb) / 4));
        __syncthreads();
        if (c0 - cb < 32) T_STAMP(s, 9 + 6 * ((c0 - cb) / 4));
        float acc[4];
        // the plane loads of the chunk after next (two chunks ahead), one plane at each of the gather's four stops
        auto later_loads = [&](int u) {
            if constexpr (Cfg::B4) {                              // one entry's 16 bytes per stop
                if (u < EPT) {
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 v = buf_ld4(rin, e.off[u] * 4u, (uint32_t)(min(c0 + 8, cmax - 3) >> 2) * (hw4 * 4u));
                    pre[u][0] = v.x; pre[u][1] = v.y; pre[u][2] = v.z; pre[u][3] = v.w;
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t soff = (uint32_t)min(c0 + 8 + u, cmax) * hw4;
#pragma unroll
            for (int j = 0; j < EPT; ++j) pre[j][u] = buf_ld(rin, e.off[j], soff);
            __builtin_amdgcn_sched_barrier(0);
        };
        Acc4 head = acc4_init<false>(0.0f);
        if constexpr (CAN_BAL) {
            if (bal == 0) {                               // (first chunk, uniform: the waves' longest lists are in LDS behind this chunk's barrier)
                uint32_t mx = 0;
#pragma unroll
                for (int w_ = 0; w_ < TT / 64; ++w_) mx = max(mx, L.wsum[8 + w_]);
                const uint32_t ntot = (uint32_t)L.off[TT - 1] + (L.cnt[TT - 1] | 1u);
                bal_s = (ntot + TT - 1u) / TT;
                bal = mx > bal_s + (uint32_t)SLR_BAL_SLACK ? 1 : 2;
                if (bal == 1) bl = bal_setup<Cfg>(L, tid, bal_s, ntot);
            }
            if (bal == 1) head = bal_walk<Cfg>(L, bl, tid, later_loads);
        }
        if (!CAN_BAL || bal != 1) gather_chunk<Cfg, MAXOP, !ADAPT ? 1 : (SLAB && NORM) ? 2 : 4>(L, g, lane, s.init, acc, later_loads);      // (the normalising sink kernel: 2 -- with 4 it spills)
        if (c0 - cb < 32) T_STAMP(s, 11 + 6 * ((c0 - cb) / 4));
        __syncthreads();        // val4 is overwritten by the next chunk (the stores below do not hold the others up); balanced: every carry is in
        if constexpr (CAN_BAL) {
            if (bal == 1) {
                bal_join<Cfg>(L, bl, tid, head);
                __syncthreads();
                const float4 r4 = L.bsum[tid];            // (written again after the next chunk's first barrier)
                acc[0] = r4.x; acc[1] = r4.y; acc[2] = r4.z; acc[3] = r4.w;
            }
        }
        if (c0 - cb < 32) T_STAMP(s, 13 + 6 * ((c0 - cb) / 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (FULL || c0 + u < ce) {                // (scalar: only the last chunk of a plane count that is not a multiple of 4)
                const uint32_t soff = (uint32_t)(c0 + u - cbase) * obytes;
                float r = acc[u];
                if (ACCUM && !first) { const float o = buf_ld(rout, voff, soff); r = MAXOP ? fmaxf(r, o) : r + o; }   // earlier passes of this piece
                if (NORM && !SLAB && (!ACCUM || last)) r *= inv;
                buf_st<SLAB ? BUF_SC1 : SLR_STORE_AUX>(rout, voff, soff, r);      // (a slab is read by another workgroup)
            }
        }
    };
    // (the loads of the first two chunks were issued in phase 1a, microseconds ago: waiting for them here costs nothing, and with
    //  nothing pending at the loop's entry the compiler's counter bookkeeping inside the loop is exact -- merged with a non-empty entry
    //  state it made every other chunk wait for the loads issued ONE chunk earlier: 1.84 against 1.59 us per chunk)
    __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
    int c0 = cb;
    for (; c0 + 8 <= ce; c0 += 8) {
        chunk(std::true_type{}, preA, c0);
        chunk(std::true_type{}, preB, c0 + 4);
    }
    if (c0 < ce) {                                    // the last 1 .. 7 planes
        chunk(std::false_type{}, preA, c0);
        if (c0 + 4 < ce) chunk(std::false_type{}, preB, c0 + 4);
    }
}

}  // namespace slr
