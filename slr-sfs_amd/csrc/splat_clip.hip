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
//   per batch  clip_tile_kernel     one workgroup = one piece of one frame (up to 8 frames per launch, their block groups
//                                   interleaved so the same tile of consecutive frames shares an XCD's L2): the two row-segment
//                                   lists -> the flow rows (coalesced) -> entries in LDS -> per-output-pixel records -> chunk pipeline;
//              clip_tile_kernel<.., PASSES>  a piece that still holds more than SEG entries (an octant that is a sink by itself; any
//                                   pathological flow) was appended to the frame's deferred list by its workgroup; this normally
//                                   empty launch walks such pieces pass by pass.
// Round 3 did this with per-pixel bins (two passes over all maps with one atomic per footprint: 29 us per frame amortised), partial
// tiles for multi-segment tiles and a combine pass (17 us per frame): stage 230 us per frame around a 180 us kernel.
#include "splat_core.hpp"

#include <type_traits>

namespace slr {

constexpr int CT = TILE_PIX;                       // work-items per workgroup = output pixels of a tile
constexpr int C_EPT = SLR_EPT_TWO;                 // entries per work-item
constexpr int C_SEG = C_EPT * CT;                  // entries a workgroup stages at once
constexpr int C_CHUNK = 4;                         // planes per pass of the chunk pipeline
constexpr int C_KREG = SLR_KREG_TWO;               // records of an output pixel kept in registers across the chunks
constexpr int C_RECCAP = 4 * C_SEG + CT;           // <= 4 records per entry + one pad per pixel (odd list lengths: bank spreading)
constexpr int C_MAXB = SLR_CLIP_MAXB;              // frames per launch
constexpr uint32_t C_NULL = C_SEG;                 // staged-entry index of the all-zero slot
constexpr int C_XCD = SLR_XCD_GROUP;
constexpr uint32_t CLIP_TOTALS = 8;                // per frame: [0] items, [4] deferred pieces, [5] arrivals of the deferred launch
constexpr uint32_t C_DEFER_WG = 16;                // workgroups per frame of the deferred launch
static_assert(CT == 2 * ROW_CAP, "rows_setup loads the two row lists with one work-item per slot");
static_assert(2 * (2 * ROW_CAP) * 4 + C_SEG * 8 <= C_RECCAP * 8, "the row lists and the second group's entry words live in the record area");

// ---- kernel arguments -------------------------------------------------------------------------------------------------------------
struct ClipShared {                // the same for every frame of a launch
    const float *in;               // [C,H,W] value planes (the encoder's features)
    const float *mul;              // [H,W] weight logits Z
    const float *mulmax;           // device scalar subtracted before exp, or nullptr
    const float *in2, *mul2;       // second weight group: one value plane with its own weight logits (G2 instantiations)
    int C, H, W, tiles_x, tiles;
    int mulmode, mulmode2, norm_mode;
    float eps;
    long long *trace;              // development builds (-DSLR_TRACE): 64 time stamps per workgroup, or nullptr
};
struct ClipFrame {
    const float *flow[2];                  // the frame's forward / backward displacement map [2,H,W]
    const unsigned long long *rowcnt[2];   // [tiles][4]  (entries << 32 | row segments), octant histogram (2 words), pad
    const RowRec *rowlist[2];              // [tiles][ROW_CAP]
    const ItemDesc *items;                 // the frame's work items (rows_plan_clip_kernel)
    uint32_t *totals, *defer;              // [CLIP_TOTALS], [items_cap]
    float *out, *out2, *norm_out;          // [C,H,W], [H,W] (G2), [H,W] or nullptr
    float scale[2];                        // alpha, 1 - alpha
    uint32_t grid, pad_;                   // blocks of this frame (multiple of 8 * C_XCD)
};
struct ClipBatch {
    ClipShared s;
    ClipFrame f[C_MAXB];
    uint32_t nb, interleave;
};
static_assert(sizeof(ClipBatch) <= 4096, "kernel arguments are limited to 4 KiB");

enum { MUL_ONE = 0, MUL_PLANE = 1, MUL_EXP = 2, MUL_EXP_SHIFT = 3 };

#ifdef SLR_TRACE      // development aid: per-workgroup phase time stamps (shader clock) into ClipShared.trace (tools/dev/trace_clip.py)
#define C_STAMP(s_, slot) do { if ((s_).trace && threadIdx.x == 0) (s_).trace[(size_t)blockIdx.x * 64 + (slot)] = clock64(); } while (0)
#define C_NOTE(s_, slot, v) do { if ((s_).trace && threadIdx.x == 0) (s_).trace[(size_t)blockIdx.x * 64 + (slot)] = (long long)(v); } while (0)
#else
#define C_STAMP(s_, slot) do { } while (0)
#define C_NOTE(s_, slot, v) do { } while (0)
#endif

// ---- LDS ------------------------------------------------------------------------------------------------------------------------------
// counts | wave sums | misc | offsets | records | staged values (+ the all-zero slot).  Aliases: the row lists sit in the record
// area (dead before the records are written), the entry arrays in the staging area (dead before the first chunk is staged).
struct ClipLds {
    uint32_t *cnt, *wsum, *misc;
    uint16_t *off;
    uint2 *rec;
    float4 *val4;
    uint32_t *rl_sy, *rl_w1;       // the two row lists, compacted (direction 0, then direction 1): image row | octants, packed column / slot / hits
    float4 *ent4;                  // entries: source pixel | direction << 31 (bits), target X, target Y, weight logit
    float2 *ent2;                  // G2: second group's weight logit and value of the entry
};
constexpr size_t C_LDS_HEAD = (size_t)(CT + 16 + 16 + CT / 2) * 4;
constexpr size_t C_LDS_BYTES = C_LDS_HEAD + (size_t)C_RECCAP * 8 + (size_t)(C_SEG + 1) * 16;
static_assert(C_LDS_HEAD % 16 == 0, "records and staged values are 16-byte aligned");

__device__ __forceinline__ ClipLds clip_lds(uint32_t *smem) {
    ClipLds L;
    L.cnt = smem;
    L.wsum = smem + CT;
    L.misc = smem + CT + 16;
    L.off = reinterpret_cast<uint16_t *>(smem + CT + 32);
    L.rec = reinterpret_cast<uint2 *>(smem + CT + 32 + CT / 2);
    L.val4 = reinterpret_cast<float4 *>(L.rec + C_RECCAP);
    L.rl_sy = reinterpret_cast<uint32_t *>(L.rec);
    L.rl_w1 = L.rl_sy + 2 * ROW_CAP;
    L.ent4 = L.val4;
    L.ent2 = reinterpret_cast<float2 *>(L.rl_w1 + 2 * ROW_CAP);
    return L;
}

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
    constexpr int R = 2;
    const uint32_t m = blockIdx.y, d = m >= r.nframes ? 1u : 0u, fi = m - d * r.nframes;
    const float *fl = r.disp[d] + (size_t)r.idx[d][fi] * 2 * H * W;
    const int bl = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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
    unsigned long long my_lo = 0, my_hi = 0;       // hits per column octant of the tile, 16 bits each (octants 0-3 | 4-7)
    int k = 0;
    auto flush = [&]() {
        if (my_tile >= 0) {
            unsigned long long *w = cnt_m + 4 * (size_t)my_tile;
            const unsigned long long old = atomicAdd(w, 1ull | ((unsigned long long)my_cnt << 32));
            if (my_lo) atomicAdd(w + 1, my_lo);                  // (no return value: fire and forget)
            if (my_hi) atomicAdd(w + 2, my_hi);
            const uint32_t slot = (uint32_t)old;
            if (slot < (uint32_t)ROW_CAP) list_m[(size_t)my_tile * ROW_CAP + slot] = RowRec{(uint32_t)my_y, ((uint32_t)stx << 8) | my_cnt};
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
            const Corners c = make_corners(fx[rr], fy[rr], x, y);
            const TileSet q = footprint_tiles(c, H, W);
            if (q.vxa & q.vya) t0 = q.tya * tiles_x + q.txa;
            if (q.vxb & q.vya) t1 = q.tya * tiles_x + q.txb;
            if (q.vxa & q.vyb) t2 = q.tyb * tiles_x + q.txa;
            if (q.vxb & q.vyb) t3 = q.tyb * tiles_x + q.txb;
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
            // exact hits per column octant: 8 ballots for a full append; the one-column overlaps into a neighbouring tile (fewer than 8
            // hits: half of all appends) walk their <= 7 lanes with scalar operations instead (the kernel is VALU-bound)
            uint32_t rm = 0;
            unsigned long long lo = 0, hi = 0;
            if (c >= 8u) {                                   // (wave-uniform)
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const unsigned long long co = (unsigned long long)__popcll(__ballot((lm >> o) & 1u));
                    rm |= co ? 1u << o : 0u;
                    if (o < 4) lo |= co << (16 * o); else hi |= co << (16 * (o - 4));
                }
            } else {
                for (unsigned long long mk = __ballot(h); mk; mk &= mk - 1ull) {
                    const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lm, __ffsll((long long)mk) - 1);
                    rm |= l;
                    // bit o of l -> +1 in the 16-bit field of octant o
                    lo += (unsigned long long)(l & 1u) | ((unsigned long long)(l & 2u) << 15) | ((unsigned long long)(l & 4u) << 30) | ((unsigned long long)(l & 8u) << 45);
                    hi += (unsigned long long)((l >> 4) & 1u) | ((unsigned long long)((l >> 4) & 2u) << 15) | ((unsigned long long)((l >> 4) & 4u) << 30) | ((unsigned long long)((l >> 4) & 8u) << 45);
                }
            }
            if (t0 == T) t0 = -1;
            if (t1 == T) t1 = -1;
            if (t2 == T) t2 = -1;
            if (t3 == T) t3 = -1;
            if (lane == k) { my_tile = T; my_cnt = c; my_y = y | (int)(rm << 24); my_lo = lo; my_hi = hi; }
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
constexpr uint32_t ROWW_STX = 18, ROWW_BASE = 7;
__global__ __launch_bounds__(CT) void rows_sort_clip_kernel(ClipRows r) {
    __shared__ uint32_t k_sy[CT / 64][ROW_CAP], k_sx[CT / 64][ROW_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * (CT / 64) + w, m = blockIdx.y;
    if (t >= r.nt) return;                                            // (whole waves; no barriers below)
    const uint32_t n = min((uint32_t)r.rowcnt[((size_t)m * r.nt + t) * 4], (uint32_t)ROW_CAP);
    RowRec *list = r.rowlist + ((size_t)m * r.nt + t) * ROW_CAP;
    constexpr int PER = ROW_CAP / 64;
    RowRec rec[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t q = (uint32_t)lane + 64u * i;
        rec[i] = q < n ? list[q] : RowRec{0xffffffffu, 0xffffffffu};
        k_sy[w][q] = rec[i].sy & 0xffffffu;
        k_sx[w][q] = rec[i].sx_cnt >> 8;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint32_t rank[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) rank[i] = 0;
    for (uint32_t j = 0; j < n; ++j) {
        const unsigned long long kj = ((unsigned long long)k_sy[w][j] << 24) | k_sx[w][j];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const unsigned long long ki = ((unsigned long long)(rec[i].sy & 0xffffffu) << 24) | (rec[i].sx_cnt >> 8);
            rank[i] += kj < ki ? 1u : 0u;                              // (keys are distinct: one append per (segment, tile))
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

__global__ __launch_bounds__(CT) void rows_plan_clip_kernel(ClipRows r, ClipPlan p, uint32_t seg) {
    __shared__ uint32_t wsum[CT / 64];
    const uint32_t fi = blockIdx.x, nt = r.nt;
    const int tid = threadIdx.x;
    const unsigned long long *w0 = r.rowcnt + (size_t)fi * nt * 4, *w1 = r.rowcnt + (size_t)(r.nframes + fi) * nt * 4;
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
        if (on) { a0 = w0[4 * (size_t)t]; a1 = w0[4 * (size_t)t + 1]; a2 = w0[4 * (size_t)t + 2];
                  b0 = w1[4 * (size_t)t]; b1 = w1[4 * (size_t)t + 1]; b2 = w1[4 * (size_t)t + 2]; }
        const uint32_t cnt = (uint32_t)(a0 >> 32) + (uint32_t)(b0 >> 32);
        if (SLR_CLIP_HEAVY && on && (cnt > heavy_thr) != (pass == 0u)) on = false;      // not this pass's tile
        unsigned long long pcs = 0x80ull;                          // pieces: (first octant | octants << 4), 8 bits each; default: octants [0, 8)
        uint32_t ns = on ? 1u : 0u;
        if (on && cnt > seg) {
            uint32_t oh[8], osum = 0;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const unsigned long long x = o < 4 ? a1 : a2, y = o < 4 ? b1 : b2;
                oh[o] = (uint32_t)((x >> (16 * (o & 3))) & 0xffffu) + (uint32_t)((y >> (16 * (o & 3))) & 0xffffu);
                osum += oh[o];
            }
            const uint32_t even = osum / ((osum + seg - 1u) / seg);                 // pieces of about equal weight, not one full + a rest
            uint32_t start = 0, sum = 0, np = 0;
            unsigned long long q = 0;
#pragma unroll
            for (uint32_t o = 0; o < 8; ++o) {
                if ((sum + oh[o] > seg || sum + oh[o] / 2u >= even) && o > start) { q |= (unsigned long long)(start | ((o - start) << 4)) << (8 * np); ++np; start = o; sum = 0; }
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

struct Piece {                     // what one workgroup works on
    uint32_t tile;
    int ty0, tx0;                  // the tile's first output row / column
    int pca, pcb;                  // the piece's output columns [pca, pcb) of the tile (tile-local)
    uint32_t cnt0, cnt1;           // exact entries of the TILE per direction (rowbin_clip_kernel)
    uint32_t len0, len1;           // row segments to walk per direction (the whole image's if the list overflowed)
    uint32_t n0;                   // list entries of direction 0 in LDS (direction 1 follows them)
    bool ovf0, ovf1;               // the direction's list overflowed ROW_CAP: every row segment of the image is scanned
    bool whole;                    // all 8 octants
};

// The two row-segment lists of the tile (sorted by rows_sort_clip_kernel) -> LDS, compacted: direction 0, then direction 1.
__device__ __forceinline__ void rows_setup(const ClipFrame &f, const ClipLds &L, const Piece &p, int tid) {
    const int d = tid >> 8, q = tid & (ROW_CAP - 1);
    const uint32_t nd = d ? (p.ovf1 ? 0u : p.len1) : p.n0;
    if ((uint32_t)q < nd) {
        const RowRec r = f.rowlist[d][(size_t)p.tile * ROW_CAP + q];
        const uint32_t pos = (d ? p.n0 : 0u) + (uint32_t)q;
        L.rl_sy[pos] = r.sy;
        L.rl_w1[pos] = r.sx_cnt;
    }
    __syncthreads();
}

// One walk over this wave's row segments (wave w takes segments w, w + 8, ... of [direction 0 ; direction 1]); returns the wave's hits.
//   MODE 0  the whole tile, no overflow, <= SEG entries: the list's hit counts give every row segment its first slot;
//   MODE 1  a column range / an overflowed list: one LDS atomic per wave and row segment hands out the slots (L.misc[0]);
//   MODE 2  ordinals (hits of the waves before + own so far, after a count pass): emits the ordinals in [lo, hi).
// EMIT: write the entries (source pixel | direction << 31, image row, flow) to the entry arrays.
template <int MODE, bool EMIT, bool G2>
__device__ __forceinline__ uint32_t rows_walk(const ClipShared &s, const ClipFrame &f, const ClipLds &L, const Piece &p, int tid,
                                              uint32_t wave_base, uint32_t lo, uint32_t hi) {
    constexpr int CB = SLR_ROW_CB_CLIP;
    const int lane = tid & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = s.H * s.W;
    const uint32_t nseg = p.len0 + p.len1;
    const uint32_t my_n = nseg > wid ? (nseg - wid + (uint32_t)(CT / 64) - 1u) / (uint32_t)(CT / 64) : 0u;
    const uint32_t range_mask = ((1u << ((p.pcb - p.pca) >> 3)) - 1u) << (p.pca >> 3);      // the piece's column octants
    // (the weight logits -- and the second group's logits and values -- of a row segment are loaded with its flow: coalesced, and the
    //  dependent gather round trip they used to be, after the entries were known, is gone from phase 1)
    struct Group { float fx[CB], fy[CB], z[CB], l2[G2 ? CB : 1], v2[G2 ? CB : 1]; int sy[CB], stx[CB]; uint32_t b0[CB], d[CB]; };
    const bool has_mul = s.mulmode != MUL_ONE;
    auto issue = [&](Group &g, uint32_t j0) {
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const uint32_t j = j0 + (uint32_t)i, ri = wid + j * (uint32_t)(CT / 64);
            bool on = j < my_n;
            const uint32_t d = (on && ri >= p.len0) ? 1u : 0u;
            const uint32_t rj = on ? ri - (d ? p.len0 : 0u) : 0u;
            int sy, stx;
            uint32_t base = 0;
            if (d ? p.ovf1 : p.ovf0) {
                sy = (int)(rj / (uint32_t)s.tiles_x);
                stx = (int)(rj - (uint32_t)sy * (uint32_t)s.tiles_x);
            } else {
                const uint32_t q = on ? (d ? p.n0 : 0u) + rj : 0u;
                const uint32_t syw = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rl_sy[q]);
                sy = (int)(syw & 0xffffffu);
                on = on && ((syw >> 24) & range_mask) != 0u;           // (a piece only loads the segments that touch its column octants)
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rl_w1[q]);
                stx = (int)(w1 >> ROWW_STX);
                if (MODE == 0) base = ((w1 >> ROWW_BASE) & 0x7ffu) + (d ? p.cnt0 : 0u);
            }
            g.sy[i] = on ? sy : -1;
            g.stx[i] = stx;
            g.b0[i] = base;
            g.d[i] = d;
            const float *fl = d ? f.flow[1] : f.flow[0];
            const int sx = stx * TILE_W + lane;
            const bool in = on & (sx < s.W);
            const uint32_t q = in ? (uint32_t)(sy * s.W + sx) : 0u;
            g.fx[i] = fl[q];
            g.fy[i] = fl[(uint32_t)HW + q];
            g.z[i] = has_mul ? s.mul[q] : 0.0f;
            if (G2) { g.l2[i] = s.mul2[q]; g.v2[i] = s.in2[q]; }
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
                    L.ent4[slot - lo] = make_float4(__uint_as_float((uint32_t)(sy * s.W + sx) | (g.d[i] << 31)), X, Y, g.z[i]);
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

// What a work-item keeps of its C_EPT entries for the chunk pipeline.
struct EntryRegs {
    uint32_t off[C_EPT];           // byte offset of the source pixel inside a plane
    float m[C_EPT];                // G2: the first group's weight of the entry (applied when its values are staged)
};

__device__ __forceinline__ void prefetch_planes(rsrc_t rin, const EntryRegs &e, float (&pre)[C_EPT][C_CHUNK], int c0, int cmax, uint32_t hw4) {
#pragma unroll
    for (int u = 0; u < C_CHUNK; ++u) {
        const uint32_t soff = (uint32_t)min(c0 + u, cmax) * hw4;           // (planes past the last re-read it)
#pragma unroll
        for (int j = 0; j < C_EPT; ++j) pre[j][u] = buf_ld(rin, e.off[j], soff);
    }
}

// Phase 1: the piece's `total` entries (in the entry arrays) -> per-output-pixel record lists.
//   1a  footprint of this work-item's entries, weight m = exp(Z - Zmax) * alpha | (1 - alpha), one LDS atomic per in-piece corner
//       reserves a slot in that output pixel's list; the plane loads of the first two chunks are issued as soon as the source
//       pixels are known;
//   1b  workgroup scan of the list lengths (padded to odd: the lanes' list walks then start on different banks);
//   1c  the (entry, weight) records are scattered into the lists.
// G2 (a second weight group shares the records): the records keep the PURE bilinear weights, m multiplies the first group's values
// when they are staged, and the entry's slot of a special chunk carries  m | in2 * m2 | m2  (gathered before the value planes).
template <bool G2>
__device__ __forceinline__ void build_records(const ClipShared &s, const ClipFrame &f, const ClipLds &L, const Piece &p, int tid,
                                              uint32_t total, rsrc_t rin, uint32_t hw4, float shift, float sc0, float sc1, EntryRegs &e,
                                              float (&preA)[C_EPT][C_CHUNK], float (&preB)[C_EPT][C_CHUNK]) {
    uint32_t dir[C_EPT];
    float X[C_EPT], Y[C_EPT], mm[C_EPT], l2[C_EPT], v2[C_EPT];
    bool val[C_EPT];
#pragma unroll
    for (int j = 0; j < C_EPT; ++j) {
        const uint32_t k = (uint32_t)tid + (uint32_t)j * CT;
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
    prefetch_planes(rin, e, preA, 0, s.C - 1, hw4);
    prefetch_planes(rin, e, preB, C_CHUNK, s.C - 1, hw4);
    C_STAMP(s, 3);
    // (G2: a work-item's special-chunk slots are the entry slots it has just read itself: no hazard)
    uint32_t ts[C_EPT][4];                            // (output pixel << 16) | slot, 0xffffffff = corner not in the piece
    float w[C_EPT][4];
#pragma unroll
    for (int j = 0; j < C_EPT; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { ts[j][k] = 0xffffffffu; w[j][k] = 0.0f; }
        if (G2 && !val[j]) L.val4[tid + j * CT] = make_float4(0.f, 0.f, 0.f, 0.f);      // (no record points here)
        if (!val[j]) continue;
        const Corners c = corners_at(X[j], Y[j]);
        const float sc = dir[j] ? sc1 : sc0;          // (two scalars read at the kernel's start: taken from the frame's arguments here, by a
                                                      //  per-lane index or select, they become a vector memory load behind the 24 plane loads)
        float m = sc;
        if (s.mulmode == MUL_PLANE) m = mm[j] * m;
        else if (s.mulmode >= MUL_EXP) m = expf(mm[j] - shift) * m;
        if (G2) {
            float m2 = sc;
            m2 = s.mulmode2 == MUL_PLANE ? l2[j] * m2 : expf(l2[j]) * m2;
            L.val4[tid + j * CT] = make_float4(m, v2[j] * m2, m2, 0.0f);
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
        // all four reservations go out before the first result is looked at, without branches: a corner outside the piece adds 0 to
        // this work-item's own counter (one LDS round trip per entry instead of four dependent ones inside exec-masked branches)
        uint32_t slot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) slot[k] = atomicAdd(&L.cnt[kb[k] ? tg[k] : tid], kb[k] ? 1u : 0u);      // ds_add_rtn_u32
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ts[j][k] = kb[k] ? ((uint32_t)tg[k] << 16) | slot[k] : 0xffffffffu;
            w[j][k] = kb[k] ? m * c.w[k] : 0.0f;
        }
    }
    C_STAMP(s, 4);
    __syncthreads();
    C_STAMP(s, 5);
    {                                                 // 1b
        const uint32_t v = L.cnt[tid] | 1u;
        const uint32_t ex = block_excl_scan(v, L.wsum, tid);
        L.off[tid] = (uint16_t)ex;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < C_EPT; ++j)                   // 1c
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ts[j][k] != 0xffffffffu)
                L.rec[L.off[ts[j][k] >> 16] + (ts[j][k] & 0xffffu)] = make_uint2((uint32_t)tid + (uint32_t)j * CT, __float_as_uint(w[j][k]));
    if (tid == 0) { L.val4[C_NULL] = make_float4(0.f, 0.f, 0.f, 0.f); L.rec[C_RECCAP - 1] = make_uint2(C_NULL, 0u); }
    __syncthreads();
}

// The record list of this work-item's output pixel for the gather: the first C_KREG records live in registers for the whole
// chunk loop (missing ones point at the all-zero slot with weight 0: fma(0, 0, acc) == acc, so the gather has no selects); what is
// left of a list far longer than the wave's average (a "sink" pixel) is walked by the whole wave, lane-strided, and wave-reduced.
struct PixelList {
    uint32_t r0, rl, r1;           // own records [r0, rl), cooperative rest [rl, r1)
    unsigned long long heavy;      // lanes of this wave whose rest the wave walks together
    uint32_t ce[C_KREG];
    float cw[C_KREG];
};
constexpr uint32_t C_NULLREC = C_RECCAP - 1;       // a record (all-zero slot, weight 0) that no list owns (the last pixel's pad)

__device__ __forceinline__ PixelList pixel_list(const ClipLds &L, int tid) {
    PixelList g;
    g.r0 = L.off[tid];
    g.r1 = g.r0 + L.cnt[tid];
    uint32_t wave_recs = g.r1 - g.r0;                  // records of this wave's 64 output pixels
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_recs += __shfl_xor(wave_recs, d);
    // own share: twice the wave's average list length (a uniformly compressed region stays per-lane), at least SLR_LMAX; what is left
    // of a longer list goes to the whole wave once it exceeds SLR_HEAVY_SLACK records (a cooperative pass costs ~50 cross-lane
    // operations per chunk, a lane walking alone ~10 per record while the other 63 wait)
    const uint32_t own = max((uint32_t)SLR_LMAX, 2u * ((wave_recs + 63u) >> 6));
    g.rl = (g.r1 - g.r0 >= own + (uint32_t)SLR_HEAVY_SLACK) ? g.r0 + own : g.r1;
    g.heavy = __ballot(g.r1 > g.rl);
    // (the cooperative passes run one after the other; with many long lists in one wave every lane walks its own)
    if (__popcll(g.heavy) > SLR_HEAVY_MAX) { g.rl = g.r1; g.heavy = 0ull; }
#pragma unroll
    for (int k = 0; k < C_KREG; ++k) {
        const uint32_t qi = g.r0 + (uint32_t)k < g.rl ? g.r0 + (uint32_t)k : C_NULLREC;
        const uint2 q = L.rec[qi];
        g.ce[k] = q.x;
        g.cw[k] = __uint_as_float(q.y);
    }
    return g;
}

// acc[u] = sum over the pixel's records of staged value[u] * weight, for the C_CHUNK planes staged in LDS.
// between(k), k = 0..3: called at four points of the gather -- the chunk pipeline issues the plane loads of a later chunk there, a few
// at a time: the waves of a workgroup run in step, and 12 loads per wave issued in one burst wait for the texture addresser (0.28 us
// per chunk, measured) while the LDS pipe idles, then the LDS reads of the gather queue up while the addresser idles.
template <typename F>
__device__ __forceinline__ void gather_chunk(const ClipLds &L, const PixelList &g, int lane, float (&acc)[C_CHUNK], F &&between) {
    constexpr int RB = 4;
    {
        between(0);
        float4 v[C_KREG];
#pragma unroll
        for (int k = 0; k < C_KREG; ++k) v[k] = L.val4[g.ce[k]];                  // ds_read_b128: 4 planes per LDS instruction
        between(1);
        acc[0] = v[0].x * g.cw[0]; acc[1] = v[0].y * g.cw[0]; acc[2] = v[0].z * g.cw[0]; acc[3] = v[0].w * g.cw[0];
#pragma unroll
        for (int k = 1; k < C_KREG; ++k) {
            acc[0] = __builtin_fmaf(v[k].x, g.cw[k], acc[0]); acc[1] = __builtin_fmaf(v[k].y, g.cw[k], acc[1]);
            acc[2] = __builtin_fmaf(v[k].z, g.cw[k], acc[2]); acc[3] = __builtin_fmaf(v[k].w, g.cw[k], acc[3]);
        }
    }
    between(2);
    for (uint32_t r = g.r0 + (uint32_t)C_KREG; r < g.rl; r += RB) {
        uint2 q[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) q[k] = L.rec[r + (uint32_t)k < g.rl ? r + (uint32_t)k : C_NULLREC];
        float4 v[RB];
#pragma unroll
        for (int k = 0; k < RB; ++k) v[k] = L.val4[q[k].x];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            const float w = __uint_as_float(q[k].y);
            acc[0] = __builtin_fmaf(v[k].x, w, acc[0]); acc[1] = __builtin_fmaf(v[k].y, w, acc[1]);
            acc[2] = __builtin_fmaf(v[k].z, w, acc[2]); acc[3] = __builtin_fmaf(v[k].w, w, acc[3]);
        }
    }
    between(3);
    for (unsigned long long hv = g.heavy; hv; hv &= hv - 1) {                     // long lists, cooperatively
        const int src = __ffsll((long long)hv) - 1;
        const uint32_t hb = __shfl(g.rl, src), he = __shfl(g.r1, src);
        float part[C_CHUNK] = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t r = hb + (uint32_t)lane; r < he; r += 64) {
            const uint2 q = L.rec[r];
            const float4 x = L.val4[q.x];
            const float w = __uint_as_float(q.y);
            part[0] = __builtin_fmaf(x.x, w, part[0]); part[1] = __builtin_fmaf(x.y, w, part[1]);
            part[2] = __builtin_fmaf(x.z, w, part[2]); part[3] = __builtin_fmaf(x.w, w, part[3]);
        }
#pragma unroll
        for (int u = 0; u < C_CHUNK; ++u) {
            const float t = wave_sum(part[u]);
            if (lane == src) acc[u] += t;
        }
    }
}

// sum of the pixel's record weights (the normaliser when the weights carry m)
__device__ __forceinline__ float weight_sum(const ClipLds &L, const PixelList &g, int lane) {
    float nrm = 0.0f;
    for (uint32_t r = g.r0; r < g.rl; ++r) nrm += __uint_as_float(L.rec[r].y);
    for (unsigned long long hv = g.heavy; hv; hv &= hv - 1) {
        const int src = __ffsll((long long)hv) - 1;
        const uint32_t hb = __shfl(g.rl, src), he = __shfl(g.r1, src);
        float part = 0.0f;
        for (uint32_t r = hb + (uint32_t)lane; r < he; r += 64) part += __uint_as_float(L.rec[r].y);
        part = wave_sum(part);
        if (lane == src) nrm += part;
    }
    return nrm;
}

// What a work-item carries from pass to pass of a deferred piece (one pass otherwise).
struct PixelSums { float nrm, g2_sum, g2_nrm; };

// Phase 2 for one pass: [special chunk] -> chunk pipeline over the C planes.  first / last: the pass is the piece's first / last one
// (a deferred piece accumulates through its own earlier stores and normalises in the last pass).
template <bool G2, bool PASSES>
__device__ __forceinline__ void stream_planes(const ClipShared &s, const ClipFrame &f, const ClipLds &L, const Piece &p, int tid,
                                              rsrc_t rin, uint32_t hw4, const EntryRegs &e, float (&preA)[C_EPT][C_CHUNK],
                                              float (&preB)[C_EPT][C_CHUNK], PixelSums &sums, bool first, bool last) {
    const int lane = tid & 63;
    const PixelList g = pixel_list(L, tid);
    const int ly = tid / TILE_W, lx = p.pca + tid - ly * TILE_W;
    const int oy = p.ty0 + ly, ox = p.tx0 + lx;
    const bool inside = (oy < s.H) & (ox < s.W) & (lx < p.pcb);
    const uint32_t opix = (uint32_t)(oy * s.W + ox);
    const uint32_t voff = inside ? opix * 4u : BUF_OOB;                   // (work-items outside the image / the piece: stores dropped)
    const rsrc_t rout = make_rsrc(f.out, (uint32_t)s.C * hw4);
    if (G2) {
        // the special chunk (m | in2 * m2 | m2 per entry, staged in phase 1a): both normalisers and the second group's sum
        float a2[C_CHUNK];
        gather_chunk(L, g, lane, a2, [](int) {});
        sums.nrm += a2[0]; sums.g2_sum += a2[1]; sums.g2_nrm += a2[2];
        if (last && inside) f.out2[opix] = sums.g2_sum / norm_divisor(sums.g2_nrm, s.norm_mode, s.eps);
        __syncthreads();                              // val4 is overwritten by the first value chunk
    } else {
        sums.nrm += weight_sum(L, g, lane);
    }
    if (last && inside && f.norm_out) f.norm_out[opix] = norm_divisor(sums.nrm, s.norm_mode, s.eps);
    const float inv = 1.0f / norm_divisor(sums.nrm, s.norm_mode, s.eps);           // ONE division per output pixel
    C_STAMP(s, 7);
    C_NOTE(s, 63, g.r1 - g.r0);
    const int cmax = s.C - 1;
    // FULL: all C_CHUNK planes exist -- every load and store of the body is unconditional, so the compiler knows how many memory
    // operations are younger than the ones it has to wait for and emits s_waitcnt vmcnt(N) with N > 0 (a conditional store anywhere in
    // the loop makes it drain the whole queue at the top of every chunk: the prefetch distance of two chunks becomes one)
    auto chunk = [&](auto full_tag, float (&pre)[C_EPT][C_CHUNK], int c0) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < C_EPT; ++j)
            L.val4[tid + j * CT] = G2 ? make_float4(pre[j][0] * e.m[j], pre[j][1] * e.m[j], pre[j][2] * e.m[j], pre[j][3] * e.m[j])
                                      : make_float4(pre[j][0], pre[j][1], pre[j][2], pre[j][3]);
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 8 + 6 * (c0 / C_CHUNK));
        __syncthreads();
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 9 + 6 * (c0 / C_CHUNK));
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 10 + 6 * (c0 / C_CHUNK));
        float acc[C_CHUNK];
        // the plane loads of the chunk after next (two chunks ahead), one plane at each of the gather's four stops
        gather_chunk(L, g, lane, acc, [&](int u) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t soff = (uint32_t)min(c0 + 2 * C_CHUNK + u, cmax) * hw4;
#pragma unroll
            for (int j = 0; j < C_EPT; ++j) pre[j][u] = buf_ld(rin, e.off[j], soff);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 11 + 6 * (c0 / C_CHUNK));
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 12 + 6 * (c0 / C_CHUNK));
        __syncthreads();                              // val4 is overwritten by the next chunk (the stores below do not hold the others up)
        if (c0 < 8 * C_CHUNK) C_STAMP(s, 13 + 6 * (c0 / C_CHUNK));
#pragma unroll
        for (int u = 0; u < C_CHUNK; ++u) {
            if (FULL || c0 + u < s.C) {               // (scalar: only the last chunk of a plane count that is not a multiple of 4)
                const uint32_t soff = (uint32_t)(c0 + u) * hw4;
                float r = acc[u];
                if (PASSES && !first) r += buf_ld(rout, voff, soff);      // earlier passes of this piece
                if (!PASSES || last) r *= inv;
                buf_st(rout, voff, soff, r);
            }
        }
    };
    // (the loads of the first two chunks were issued in phase 1a, ~5 us ago: waiting for them here costs nothing, and with nothing
    //  pending at the loop's entry the compiler's counter bookkeeping inside the loop is exact -- merged with a non-empty entry state it
    //  made every other chunk wait for the loads issued ONE chunk earlier: 1.84 against 1.59 us per chunk)
    __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
    int c0 = 0;
    for (; c0 + 2 * C_CHUNK <= s.C; c0 += 2 * C_CHUNK) {
        chunk(std::true_type{}, preA, c0);
        chunk(std::true_type{}, preB, c0 + C_CHUNK);
    }
    if (c0 < s.C) {                                   // the last 1 .. 7 planes
        chunk(std::false_type{}, preA, c0);
        if (c0 + C_CHUNK < s.C) chunk(std::false_type{}, preB, c0 + C_CHUNK);
    }
}

__device__ __forceinline__ Piece make_piece(const ClipShared &s, const ItemDesc &it) {
    Piece p;
    p.tile = it.tile;
    p.ty0 = (int)(it.tile / (uint32_t)s.tiles_x) * TILE_H;
    p.tx0 = (int)(it.tile % (uint32_t)s.tiles_x) * TILE_W;
    const int noct = (int)min(max(it.nseg, 1u), 8u);
    p.pca = 8 * (int)min(it.seg, 7u);
    p.pcb = min(p.pca + 8 * noct, TILE_W);
    p.whole = p.pca == 0 && p.pcb == TILE_W;
    p.cnt0 = it.cnt0; p.cnt1 = it.cnt1;
    p.ovf0 = it.off0 > (uint32_t)ROW_CAP; p.ovf1 = it.off1 > (uint32_t)ROW_CAP;
    const uint32_t all = (uint32_t)s.H * (uint32_t)s.tiles_x;
    p.len0 = p.ovf0 ? all : it.off0; p.len1 = p.ovf1 ? all : it.off1;
    p.n0 = p.ovf0 ? 0u : it.off0;
    return p;
}

// grid: per frame a multiple of 8 * C_XCD blocks, the frames' groups interleaved (see launch); CT work-items; C_LDS_BYTES of LDS.
// PASSES = false: one piece per workgroup, no loops over work; a piece of more than SEG entries goes to the frame's deferred list.
// PASSES = true:  C_DEFER_WG workgroups per frame walk the deferred lists pass by pass.
template <bool G2, bool PASSES>
__global__ __launch_bounds__(CT, PASSES ? 1 : 4) void clip_tile_kernel(ClipBatch b) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const ClipLds L = clip_lds(smem);
    const ClipShared &s = b.s;
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
    const ClipFrame &f = b.f[bf];
    const int tid = threadIdx.x;
    const uint32_t hw4 = (uint32_t)(s.H * s.W) * 4u;
    const rsrc_t rin = make_rsrc(s.in, (uint32_t)s.C * hw4);
    const float shift = (s.mulmode == MUL_EXP_SHIFT) ? s.mulmax[0] : 0.0f;      // (a dependent scalar load: issued first, needed in phase 1a)
    const float sc0 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f.scale[0])));
    const float sc1 = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f.scale[1])));
    if (!PASSES) {
        if (bx >= f.grid) return;                          // this frame has fewer groups than the longest of the batch
        // Workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Groups of C_XCD consecutive items (= neighbouring
        // tiles / pieces) are placed on the same XCD: a tile's column halo is served by that XCD's L2.
        const uint32_t slot = bx >> 3;
        const uint32_t item = ((slot / C_XCD) * 8u + (bx & 7u)) * C_XCD + slot % C_XCD;
        if (item >= f.totals[0]) return;
        const Piece p = make_piece(s, f.items[item]);
        C_STAMP(s, 0);
        L.cnt[tid] = 0;
        if (tid == 0) L.misc[0] = 0;
        rows_setup(f, L, p, tid);
        C_STAMP(s, 1);
        uint32_t total;
        if (p.whole && !p.ovf0 && !p.ovf1 && p.cnt0 + p.cnt1 <= (uint32_t)C_SEG) {
            rows_walk<0, true, G2>(s, f, L, p, tid, 0u, 0u, (uint32_t)C_SEG);
            total = p.cnt0 + p.cnt1;
            __syncthreads();
        } else {
            rows_walk<1, true, G2>(s, f, L, p, tid, 0u, 0u, (uint32_t)C_SEG);
            __syncthreads();
            total = L.misc[0];
            if (total > (uint32_t)C_SEG) {                 // (uniform) more than one pass: the pass-by-pass launch takes the piece
                if (tid == 0) f.defer[atomicAdd(f.totals + 4, 1u)] = item;
                return;
            }
        }
        C_STAMP(s, 2);
        C_NOTE(s, 60, total); C_NOTE(s, 61, p.len0 + p.len1); C_NOTE(s, 62, p.pcb - p.pca);
        EntryRegs e;
        float preA[C_EPT][C_CHUNK], preB[C_EPT][C_CHUNK];
        build_records<G2>(s, f, L, p, tid, total, rin, hw4, shift, sc0, sc1, e, preA, preB);
        C_STAMP(s, 6);
        PixelSums sums = {0.0f, 0.0f, 0.0f};
        stream_planes<G2, false>(s, f, L, p, tid, rin, hw4, e, preA, preB, sums, true, true);
        C_STAMP(s, 59);
    } else {
        const uint32_t ndef = f.totals[4];
        for (uint32_t k = bx; k < ndef; k += C_DEFER_WG) {
            const Piece p = make_piece(s, f.items[f.defer[k]]);
            rows_setup(f, L, p, tid);
            // ordinals: a count pass (hits per wave), then pass si emits the ordinals [si * SEG, (si + 1) * SEG)
            const uint32_t wc = rows_walk<2, false, G2>(s, f, L, p, tid, 0u, 0u, 0u);
            if ((tid & 63) == 0) L.misc[1 + (tid >> 6)] = wc;
            __syncthreads();
            uint32_t all = 0, wb = 0;
#pragma unroll
            for (int w = 0; w < CT / 64; ++w) { const uint32_t c = L.misc[1 + w]; all += c; wb += w < (tid >> 6) ? c : 0u; }
            const uint32_t npass = max(1u, (all + (uint32_t)C_SEG - 1u) / (uint32_t)C_SEG);
            PixelSums sums = {0.0f, 0.0f, 0.0f};
            for (uint32_t si = 0; si < npass; ++si) {
                __syncthreads();
                if (si > 0) rows_setup(f, L, p, tid);      // (the lists share LDS with the previous pass's records)
                L.cnt[tid] = 0;
                const uint32_t lo = si * (uint32_t)C_SEG;
                rows_walk<2, true, G2>(s, f, L, p, tid, wb, lo, lo + (uint32_t)C_SEG);
                __syncthreads();
                EntryRegs e;
                float preA[C_EPT][C_CHUNK], preB[C_EPT][C_CHUNK];
                build_records<G2>(s, f, L, p, tid, min((uint32_t)C_SEG, all - lo), rin, hw4, shift, sc0, sc1, e, preA, preB);
                stream_planes<G2, true>(s, f, L, p, tid, rin, hw4, e, preA, preB, sums, si == 0, si + 1 == npass);
            }
            __syncthreads();
        }
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
    // image rows travel in 24 bits of a row-list entry; a plane stack is addressed through one buffer descriptor (< 2^31 bytes)
    if (nframes <= 0 || nframes > 16384 || H <= 0 || W <= 0 || H >= (1 << 24) || (long long)H * W >= (1LL << 28) ||
        (long long)(C > 0 ? C : 1) * H * W * 4 >= (1LL << 31)) {
        set_error("%s: bad sizes nframes=%d C=%d H=%d W=%d (C*H*W*4 must stay below 2^31)", who, nframes, C, H, W);
        return SLR_E_BADARG;
    }
    return 0;
}

template <bool G2, bool PASSES>
static int launch_clip_kernel(const ClipBatch &b, uint32_t grid, hipStream_t st) {
    // > 64 KiB of dynamic LDS needs an explicit opt-in, once per device
    static bool attr_set[64] = {};
    int dev = 0;
    SLR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        SLR_CHECK_HIP(hipFuncSetAttribute((const void *)clip_tile_kernel<G2, PASSES>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((clip_tile_kernel<G2, PASSES>), dim3(grid), dim3(CT), C_LDS_BYTES, st, b);
    return 0;
}

extern thread_local void *g_ev_start, *g_ev_stop;          // slr_splat_time_next (splat.hip)
#ifdef SLR_TRACE
extern long long *g_trace;                                 // slr_debug_trace (splat.hip)
#endif

}  // namespace slr

using namespace slr;

SLR_EXPORT size_t slr_clip_plan_bytes(int nframes, int H, int W) {
    if (nframes <= 0 || nframes > 16384 || H <= 0 || W <= 0 || H >= (1 << 24) || (long long)H * W >= (1LL << 28)) return 0;
    return clip_layout(nframes, H, W).total;
}

SLR_EXPORT size_t slr_splat_scratch_bytes(int C, int H, int W) {
    return (C <= 0 || H <= 0 || W <= 0) ? 0 : 256;          // (the clip kernels need no scratch since the rows front end: ABI kept)
}

SLR_EXPORT size_t slr_splat_scratch_bytes_batch(int C, int H, int W, int nb) {
    return (C <= 0 || H <= 0 || W <= 0 || nb <= 0 || nb > C_MAXB) ? 0 : 256;
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
    const dim3 grid((unsigned)(L.tiles_x * ((L.tiles_y + 1) / 2)), L.nmaps);      // (blockIdx.y carries the map: <= 32768 maps, see clip_check)
    hipLaunchKernelGGL(rowbin_clip_kernel, grid, dim3(CT), 0, st, r, H, W, L.tiles_x, L.tiles_y);
    hipLaunchKernelGGL(rows_sort_clip_kernel, dim3((L.nt + CT / 64 - 1) / (CT / 64), L.nmaps), dim3(CT), 0, st, r);
    hipLaunchKernelGGL(rows_plan_clip_kernel, dim3(L.nframes), dim3(CT), 0, st, r, p, (uint32_t)C_SEG);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int synth_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                            const float *values2, const float *wlogit2, int exp_weights2, float *const *out2,
                            const float *const *disp_f, const float *const *disp_p, const float *alpha,
                            float *const *out, float *const *norm_out, int C, int H, int W, float eps,
                            const void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                            const int *hints, void *stream) {
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
    const char *pb = (const char *)plan;
    ClipBatch b = {};
    b.s.in = values; b.s.mul = wlogit; b.s.mulmax = wmax; b.s.in2 = values2; b.s.mul2 = wlogit2;
    b.s.C = C; b.s.H = H; b.s.W = W; b.s.tiles_x = L.tiles_x; b.s.tiles = L.tiles;
    b.s.mulmode = wmax ? MUL_EXP_SHIFT : (exp_weights ? MUL_EXP : MUL_PLANE);
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
        SLR_CHECK_ARG(disp_f[k] && disp_p[k] && out[k], "null pointer");
        ClipFrame &f = b.f[k];
        const size_t i = (size_t)frame[k];
        f.flow[0] = disp_f[k]; f.flow[1] = disp_p[k];
        f.rowcnt[0] = (const unsigned long long *)(pb + L.off_rowcnt) + i * L.nt * 4;
        f.rowcnt[1] = (const unsigned long long *)(pb + L.off_rowcnt) + ((size_t)L.nframes + i) * L.nt * 4;
        f.rowlist[0] = (const RowRec *)(pb + L.off_rowlist) + i * L.nt * ROW_CAP;
        f.rowlist[1] = (const RowRec *)(pb + L.off_rowlist) + ((size_t)L.nframes + i) * L.nt * ROW_CAP;
        f.items = (const ItemDesc *)(pb + L.off_items) + i * L.items_cap;
        f.totals = (uint32_t *)(const_cast<char *>(pb) + L.off_totals) + i * CLIP_TOTALS;
        f.defer = (uint32_t *)(const_cast<char *>(pb) + L.off_defer) + i * L.items_cap;
        f.out = out[k]; f.norm_out = norm_out ? norm_out[k] : nullptr;
        if (values2) { SLR_CHECK_ARG(out2[k], "null pointer"); f.out2 = out2[k]; }
        f.scale[0] = alpha[k]; f.scale[1] = 1.0f - alpha[k];
        // what the host knows of the plan (read back once per clip); unknown: the grid covers the bound, surplus workgroups exit at once
        const int ni = hints ? hints[3 * k] : -1;
        const uint32_t cover = ni >= 0 && (uint32_t)ni < L.items_cap ? (uint32_t)ni : L.items_cap;
        f.grid = ((cover + 8 * C_XCD - 1) / (8 * C_XCD)) * 8 * C_XCD;
        gsum += f.grid;
        gmax = f.grid > gmax ? f.grid : gmax;
    }
    // Frames of a batch are consecutive frames of a clip: tile T of frame k+1 gathers from almost the same source region as tile T
    // of frame k.  With the groups of 8 * C_XCD blocks dealt round-robin over the frames they run side by side on the same XCD and
    // share its L2 (frames with fewer groups leave a few empty blocks).
    const uint32_t grid = b.interleave ? gmax * b.nb : gsum;
    if (g_ev_start) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_start, st));        // slr_splat_time_next: the dominant kernel only
    if (grid) {
        if (values2) { if (int e = launch_clip_kernel<true, false>(b, grid, st)) return e; }
        else if (int e = launch_clip_kernel<false, false>(b, grid, st)) return e;
    }
    if (g_ev_stop) SLR_CHECK_HIP(hipEventRecord((hipEvent_t)g_ev_stop, st));
    g_ev_start = g_ev_stop = nullptr;
    // pieces of more than SEG entries (none for ordinary flows): pass by pass
    if (values2) { if (int e = launch_clip_kernel<true, true>(b, b.nb * C_DEFER_WG, st)) return e; }
    else if (int e = launch_clip_kernel<false, true>(b, b.nb * C_DEFER_WG, st)) return e;
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_synth_group_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                          const float *const *disp_f, const float *const *disp_p, const float *alpha,
                                          float *const *out, float *const *norm_out, int C, int H, int W, float eps,
                                          const void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                                          void *scratch, size_t scratch_bytes, const int *hints, void *stream) {
    (void)scratch; (void)scratch_bytes;
    return synth_clip_batch(values, wlogit, wmax, exp_weights, nullptr, nullptr, 0, nullptr, disp_f, disp_p, alpha, out, norm_out,
                            C, H, W, eps, plan, plan_bytes, nframes, frame, nb, hints, stream);
}

SLR_EXPORT int slr_synth_two_groups_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                               const float *values2, const float *wlogit2, int exp_weights2,
                                               const float *const *disp_f, const float *const *disp_p, const float *alpha,
                                               float *const *out, float *const *out2, int C, int H, int W, float eps,
                                               const void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                                               void *scratch, size_t scratch_bytes, const int *hints, void *stream) {
    (void)scratch; (void)scratch_bytes;
    SLR_CHECK_ARG(values2 && wlogit2 && out2, "null pointer");
    return synth_clip_batch(values, wlogit, wmax, exp_weights, values2, wlogit2, exp_weights2, out2, disp_f, disp_p, alpha, out,
                            nullptr, C, H, W, eps, plan, plan_bytes, nframes, frame, nb, hints, stream);
}

SLR_EXPORT int slr_synth_group_clip(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                    const float *disp_f, const float *disp_p, float alpha, float *out, float *norm_out,
                                    int C, int H, int W, float eps, const void *plan, size_t plan_bytes, int nframes,
                                    int frame, void *scratch, size_t scratch_bytes, int n_items, int n_multi, int n_whole,
                                    void *stream) {
    const int hints[3] = {n_items, n_multi, n_whole};
    return slr_synth_group_clip_batch(values, wlogit, wmax, exp_weights, &disp_f, &disp_p, &alpha, &out,
                                      norm_out ? &norm_out : nullptr, C, H, W, eps, plan, plan_bytes, nframes, &frame, 1,
                                      scratch, scratch_bytes, hints, stream);
}
