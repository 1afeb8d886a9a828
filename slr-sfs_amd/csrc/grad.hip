// grad.hip -- backward of the summation splat and the inverse (gather) half of the
// maximum-warp-norm splat.  All pure gathers: one work-item per source PIXEL, the flow and the
// four corner weights are computed once and reused for every channel (the reference recomputes
// them per element / per thread, models/softsplat.py:204-326, 84-155).
#include "splat_core.hpp"

#include <type_traits>

namespace slr {

// gradInput[n,c,y,x] = sum_corners gradOutput[n,c,corner] * w                          (softsplat.py:204-255)
// gradFlow[n,{x,y},y,x] = sum_c in[c] * sum_corners gradOutput[c,corner] * dw/d{x,y}   (softsplat.py:257-326)
// One kernel produces either or both (GIN / GFLOW): both read gradOutput at the same four corners, so a backward
// pass that needs both gradients (features and motion trained jointly) gathers them once.  Same terms in the same
// order as the reference kernels and no FMA contraction (built with -ffp-contract=off): bit-exact with them --
//   gradInput: sum order NW, NE, SW, SE;
//   gradFlow:  the reference runs one thread per flow COMPONENT with a C-loop each; here one work-item produces both
//              components from one pass over the channels, per component ((in*gout)*dw; channels outer, corners inner).
// Channels go U at a time with all their loads issued before the first use: the gathers (4 per channel and work-item,
// on 1-6 cache lines per wave and instruction for Euler-integrated flows) are the cost, and the compiler does not hoist
// them over the gradInput stores by itself.  (Measured and not kept: the two corners of a row as ONE 4-byte-aligned
// 8-byte load -- Euler flows -10 %, identity +20 %, tools/bwdbench.py.)
#define SLR_GRAD_U 4

template <bool GIN, bool GFLOW>
__global__ __launch_bounds__(256) void grad_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                                                   const float *__restrict__ gout, float *__restrict__ gin,
                                                   float *__restrict__ gflow, int C, int H, int W) {
    const int HW = H * W;
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *f = flow + (size_t)n * 2 * HW;
    const int y = i / W, x = i - y * W;
    const Corners c = make_corners(f[i], f[HW + i], x, y);
    const bool k0 = c.ok & in_image(c.x0, c.y0, H, W), k1 = c.ok & in_image(c.x0 + 1, c.y0, H, W);
    const bool k2 = c.ok & in_image(c.x0, c.y0 + 1, H, W), k3 = c.ok & in_image(c.x0 + 1, c.y0 + 1, H, W);
    // softsplat.py:289-299 (d/dx uses the y weights and vice versa)
    const float X = (float)x + f[i], Y = (float)y + f[HW + i];
    const float ax = (float)(c.x0 + 1) - X, bx = X - (float)c.x0;
    const float ay = (float)(c.y0 + 1) - Y, by = Y - (float)c.y0;
    const float dx[4] = {(-1.0f) * ay, (+1.0f) * ay, (-1.0f) * by, (+1.0f) * by};
    const float dy[4] = {ax * (-1.0f), bx * (-1.0f), ax * (+1.0f), bx * (+1.0f)};
    // Branch-free channel loop: out-of-image corners read a valid address and their PRODUCT is replaced by +0.0, so
    // the sums have the reference's terms in the reference's order (adding +0.0 changes nothing but the sign of a -0.0)
    // and the loads of several channels overlap.
    const int o = c.y0 * W + c.x0;
    const int o0 = k0 ? o : i, o1 = k1 ? o + 1 : i, o2 = k2 ? o + W : i, o3 = k3 ? o + W + 1 : i;
    const float *ip = in + (size_t)n * C * HW;
    const float *gp = gout + (size_t)n * C * HW;
    float *op = gin + (size_t)n * C * HW;
    float gx = 0.0f, gy = 0.0f;
    // U channels per pass, all their loads issued before the first use (the compiler does not hoist them over the
    // gradInput stores by itself); channels past C in the last pass re-read channel C-1 and are not stored / summed.
    constexpr int U = SLR_GRAD_U;
    for (int ch = 0; ch < C; ch += U) {
        float a0[U], a1[U], a2[U], a3[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t po = (size_t)min(ch + u, C - 1) * HW;
            a0[u] = gp[po + o0]; a1[u] = gp[po + o1]; a2[u] = gp[po + o2]; a3[u] = gp[po + o3];
            if (GFLOW) v[u] = ip[po + i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = ch + u < C;
            if (GIN) {
                float g = 0.0f;
                g += k0 ? a0[u] * c.w[0] : 0.0f;
                g += k1 ? a1[u] * c.w[1] : 0.0f;
                g += k2 ? a2[u] * c.w[2] : 0.0f;
                g += k3 ? a3[u] * c.w[3] : 0.0f;
                if (live) op[(size_t)(ch + u) * HW + i] = g;
            }
            if (GFLOW && live) {
                const float t0 = v[u] * a0[u], t1 = v[u] * a1[u], t2 = v[u] * a2[u], t3 = v[u] * a3[u];
                gx += k0 ? t0 * dx[0] : 0.0f; gy += k0 ? t0 * dy[0] : 0.0f;
                gx += k1 ? t1 * dx[1] : 0.0f; gy += k1 ? t1 * dy[1] : 0.0f;
                gx += k2 ? t2 * dx[2] : 0.0f; gy += k2 ? t2 * dy[2] : 0.0f;
                gx += k3 ? t3 * dx[3] : 0.0f; gy += k3 ? t3 * dy[3] : 0.0f;
            }
        }
    }
    if (GFLOW) {
        gflow[(size_t)n * 2 * HW + i] = gx;
        gflow[(size_t)n * 2 * HW + HW + i] = gy;
    }
}

// ---- the same gradients with the gathers served from LDS ---------------------------------------------------------------
// On Euler-integrated flows the four corner gathers of grad_kernel land on 1-6 cache lines per wave and instruction and
// run at 1.7-2.2 TB/s (identity flow: 4.6).  A flow that is smooth over a tile keeps the destinations of an 8x64 block of
// source pixels inside a small bounding box: the workgroup (one work-item per source pixel of the block) reduces that box,
// stages the box of gradOutput for U channels with coalesced row loads (wave w takes rows w, w+8, ...), and every
// work-item gathers its four corners from LDS -- same values, same terms, same order: bit-identical to grad_kernel.
// A block whose box does not fit (incoherent or strongly stretching flow), or whose rows stay straight (the direct gathers
// are then already coalesced), takes the direct gathers instead (decided per block).
// Measured (768x1280, C = 65, tools/bwdbench.py; grad_kernel -> this kernel): Euler t=30 gradInput 254 -> 186 us, gradFlow
// 242 -> 173, both gradients 279 -> 222 (0.35 -> 0.44 of 8 TB/s on 3C planes); t=59 both 346 -> 297; identity flow
// gradInput 119 -> 129, both 165 -> 195 (the direct path carried this kernel's registers: 96 VGPRs).  Round 3, later: each path a
// loop of its own and the direct gathers double-buffered in registers (next pass in flight under this pass's sums and stores):
// 82 VGPRs; both gradients identity 188 -> 181 us, Euler t=30 217-222 -> 212-217, gradInput t=59 285 -> 260.
// Round 4: passes whose U channels all exist store unconditionally (the tail pass is code of its own: counted waits inside both loops),
// stores through a buffer descriptor, box 2048 -> 4096 floats per channel: both gradients identity 181 -> 165-171 us, Euler t=30
// 214-219 -> 190-197 (0.50-0.52 of 8 TB/s), t=59 300-320 -> 248-255; gradInput alone t=30 178-190 -> 161-168, t=59 255-274 -> 217-226.
constexpr int GT_THREADS = TILE_PIX;                  // 8 x 64 source pixels
#define SLR_GRAD_BOX 4096                              // floats of LDS per channel (e.g. 32 rows x 128 columns); x U channels x 4 bytes = 64 KiB: two workgroups
                                // per CU.  Round 4 (both gradients, identity / Euler t=30 / t=59): 2048: 169 / 201 / 290 us,
                                                       // 3072: 169 / 196 / 271, 4096: 165-171 / 190-197 / 248-255, 5120 (40 staging registers): 184 / 253 / 279;
                                                       // 2 channels per pass at 4096 / 6144 / 8192 / 10240: 170 / 189 / 274, 172 / 213 / 258, 171 / 253 / 273, 213 / 365 / 368
#define SLR_GRAD_TU 4                                  // channels per pass of the tiled kernel (round 3, 4 / 8 / 16 at box 2048 / 1536 / 1024: Euler t=30,
                                // both gradients, 223 / 245 / 301 us; grad_kernel: 279)
#define SLR_GRAD_BENT 2                                // stage through LDS only where a wave's destinations spread over more rows than this
#define SLR_GRAD_STRIPS 2                              // column strips of a block with a destination box each (1, 2, 4: power of two)
#define SLR_GRAD_WAVES 4                               // __launch_bounds__ waves per SIMD of the tiled kernel
#define SLR_GRAD_SLOTS 1024                            // small grids: channel groups while the launch stays within this many workgroups (two rounds of the 512 slots)
#define SLR_GRAD_GROUPS_MAX 4
#define SLR_GRAD_GROUPS_MIN 2                          // grids larger than the chip: two groups (group-major launch order) halve the life of the blocks a
                                // flow's sinks make slow (their tail was a seventh of the launch at Euler t=59); round 6, 65 x 768 x 1280, both gradients,
                                                       // identity / t=30 / t=59: 1 group 176 / 187 / 230 us, 2 groups 167 / 186 / 210, 4 groups 189 / 203 / 215;
                                                       // the two groups of a tile next to each other in launch order: 174 / 191 / 215
#define SLR_GRAD_BUF_LD 0                              // 1: plane loads through buffer descriptors (plane offset in an SGPR, no 64-bit vector address sums).
                                // Measured SLOWER for these gathers although the loop then has ~25 % fewer VALU instructions: both gradients
                                                       // identity / t=30 / t=59 169 / 204 / 310 us with global loads, 167 / 218 / 352 with buffer loads (box 2048)
#define SLR_GRAD_BUF_ST 1                              // gradInput stores through a buffer descriptor (out-of-image work-items dropped by the range check, no
                                // exec-mask juggling around the stores): gradInput alone t=30 192 -> 176 us, t=59 295 -> 264; both: 204 -> 208 / 310 -> 297
#define SLR_GRAD_NSE_VARIANTS 1                        // the staged loop once per count of box cells a work-item carries (1, 2, 3, 4, 6, 8) instead of always 8
                                // loads per channel: both gradients t=30 201 -> 187 us, t=59 253 -> 246, gradFlow alone t=30 166 -> 144 (round 6)
#define SLR_GRAD_PAIRS 1                               // bent blocks whose boxes do not fit: the two corners of a destination row as one 8-byte load
                                // (round 6: both gradients t=59 246 -> 227-236 us, gradFlow alone 228 -> 187; t=30 / identity have no such block)
constexpr int GT_BOX = SLR_GRAD_BOX;
struct __attribute__((packed, aligned(4))) float2u { float x, y; };   // two floats at a 4-byte-aligned address (global_load_dwordx2)

template <bool GIN, bool GFLOW>
__global__ __launch_bounds__(GT_THREADS, SLR_GRAD_WAVES) void grad_tile_kernel(const float *__restrict__ in, const float *__restrict__ flow,
                                                               const float *__restrict__ gout, float *__restrict__ gin,
                                                               float *__restrict__ gflow, int Ctot, int H, int W, int tiles_x, int cper,
                                                               float *__restrict__ gpart) {
    constexpr int U = SLR_GRAD_TU;
    // grid.z channel groups (small grids: 256 source tiles are one workgroup per CU, each walking all channels; large ones: two): group z takes channels
    // [z * cper, ...) -- below, `C` is the group's channel count and every plane pointer starts at the group's first plane; its partial
    // gradFlow sums go to gflow (group 0) / gpart[z - 1] and grad_flow_sum_kernel adds the groups up in order
    const int gz = (int)blockIdx.z;
    const int cb = gz * cper;
    const int C = min(cper, Ctot - cb);
    __shared__ float box[U][GT_BOX];
    __shared__ int red[TILE_H][SLR_GRAD_STRIPS][4];
    __shared__ int bentw[TILE_H];
    const int HW = H * W;
    const int n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int bt = (int)blockIdx.x;
    const int y = (bt / tiles_x) * TILE_H + wid, x = (bt % tiles_x) * TILE_W + lane;
    const bool live_px = (y < H) & (x < W);
    const int i = live_px ? y * W + x : 0;
    const float *f = flow + (size_t)n * 2 * HW;
    const float fxv = f[i], fyv = f[HW + i];
    const Corners c = make_corners(fxv, fyv, x, y);
    const bool k0 = live_px & c.ok & in_image(c.x0, c.y0, H, W), k1 = live_px & c.ok & in_image(c.x0 + 1, c.y0, H, W);
    const bool k2 = live_px & c.ok & in_image(c.x0, c.y0 + 1, H, W), k3 = live_px & c.ok & in_image(c.x0 + 1, c.y0 + 1, H, W);
    const float X = (float)x + fxv, Y = (float)y + fyv;
    const float ax = (float)(c.x0 + 1) - X, bx = X - (float)c.x0;
    const float ay = (float)(c.y0 + 1) - Y, by = Y - (float)c.y0;
    const float dx[4] = {(-1.0f) * ay, (+1.0f) * ay, (-1.0f) * by, (+1.0f) * by};
    const float dy[4] = {ax * (-1.0f), bx * (-1.0f), ax * (+1.0f), bx * (+1.0f)};
    // Destination boxes of the block's in-image corners, one per STRIP of GT_STRIPS column ranges of the block (64 / GT_STRIPS source
    // columns each): where the flow bends or shears the block, the strips' boxes together are much smaller than the one box around
    // everything (a sheared 8 x 64 block: one 100 x 45 box = 4500 cells does not fit, two 52 x 27 strips = 2800 do) -- round 5; the
    // cells of all strips share the GT_BOX floats per channel, strip after strip.
    constexpr int NSTR = SLR_GRAD_STRIPS, SW = TILE_W / NSTR;
    const int strip = lane / SW;
    const bool any = k0 | k1 | k2 | k3;
    int bx0 = any ? max(c.x0, 0) : 0x7fffffff, bx1 = any ? min(c.x0 + 1, W - 1) : -1;
    int by0 = any ? max(c.y0, 0) : 0x7fffffff, by1 = any ? min(c.y0 + 1, H - 1) : -1;
    int ry0 = by0, ry1 = by1;                                      // the whole row's rows (the "bent" test below)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        if (d < SW) {
            bx0 = min(bx0, __shfl_xor(bx0, d)); bx1 = max(bx1, __shfl_xor(bx1, d));
            by0 = min(by0, __shfl_xor(by0, d)); by1 = max(by1, __shfl_xor(by1, d));
        }
        ry0 = min(ry0, __shfl_xor(ry0, d)); ry1 = max(ry1, __shfl_xor(ry1, d));
    }
    if ((lane & (SW - 1)) == 0) { red[wid][strip][0] = bx0; red[wid][strip][1] = bx1; red[wid][strip][2] = by0; red[wid][strip][3] = by1; }
    if (lane == 0) bentw[wid] = ry1 - ry0 + 1;
    __syncthreads();
    // rows of gradOutput ONE wave's 64 destinations spread over: 2 for a flow that keeps rows straight (the direct gathers
    // are then as coalesced as they get: identity flow 4.7 TB/s), more where the flow bends them
    int bent = 0;
    int sx0[NSTR], sy0[NSTR], sbw[NSTR], sbase[NSTR + 1];          // per strip: box origin, width, first cell (workgroup-uniform)
    sbase[0] = 0;
    bool fits = true, some = false;
#pragma unroll
    for (int w = 0; w < TILE_H; ++w) bent = max(bent, bentw[w]);
#pragma unroll
    for (int q = 0; q < NSTR; ++q) {
        int x0 = 0x7fffffff, x1 = -1, y0 = 0x7fffffff, y1 = -1;
#pragma unroll
        for (int w = 0; w < TILE_H; ++w) { x0 = min(x0, red[w][q][0]); x1 = max(x1, red[w][q][1]); y0 = min(y0, red[w][q][2]); y1 = max(y1, red[w][q][3]); }
        const int w_ = x1 - x0 + 1, h_ = y1 - y0 + 1;              // (<= 0: nothing of this strip lands in the image)
        const bool has = w_ > 0 && h_ > 0;
        sx0[q] = has ? x0 : 0; sy0[q] = has ? y0 : 0; sbw[q] = has ? w_ : 1;
        const long long cells = has ? (long long)w_ * h_ : 0;
        fits = fits && cells <= GT_BOX;
        sbase[q + 1] = sbase[q] + (int)(cells <= GT_BOX ? cells : 0);
        some = some || has;
    }
    const int nbox = sbase[NSTR];
    const bool staged = some && fits && nbox <= GT_BOX && bent > SLR_GRAD_BENT;   // workgroup-uniform
    const int o = c.y0 * W + c.x0;
    // global offsets (direct gathers) / LDS offsets (staged); an out-of-image corner reads a valid address and its
    // PRODUCT is replaced by +0.0 (see grad_kernel)
    const int g0 = k0 ? o : i, g1 = k1 ? o + 1 : i, g2 = k2 ? o + W : i, g3 = k3 ? o + W + 1 : i;
    int mx0 = sx0[0], my0 = sy0[0], mbw = sbw[0], mbase = sbase[0];          // this work-item's strip
#pragma unroll
    for (int q = 1; q < NSTR; ++q) if (strip == q) { mx0 = sx0[q]; my0 = sy0[q]; mbw = sbw[q]; mbase = sbase[q]; }
    const int lo = mbase + (c.y0 - my0) * mbw + (c.x0 - mx0);
    const int l0 = k0 ? lo : 0, l1 = k1 ? lo + 1 : 0, l2 = k2 ? lo + mbw : 0, l3 = k3 ? lo + mbw + 1 : 0;
    // The gradInput stores go through a buffer descriptor (plane offset in an SGPR, one 32-bit pixel offset; work-items outside the image
    // are dropped by its range check); the loads stay global loads (SLR_GRAD_BUF_LD above: the same gathers through a descriptor are slower).
    const uint32_t hw4 = (uint32_t)HW * 4u;
    const size_t pbase = ((size_t)n * Ctot + cb) * HW;               // the group's first plane of this sample
    const rsrc_t rg = make_rsrc(gout + pbase, (uint32_t)C * hw4);
    const rsrc_t ri = make_rsrc(GFLOW ? in + pbase : gout, (uint32_t)C * hw4);
    const rsrc_t ro = make_rsrc(GIN ? gin + pbase : gflow, GIN ? (uint32_t)C * hw4 : 0u);
    const uint32_t vi = (uint32_t)i * 4u, vst = live_px ? vi : BUF_OOB;
    const float *gp = gout + pbase, *ip = in + pbase;
    float *op = gin + pbase;
    auto ld_g = [&](int plane, uint32_t voff) { return SLR_GRAD_BUF_LD ? buf_ld(rg, voff, (uint32_t)plane * hw4) : gp[(size_t)plane * HW + (voff >> 2)]; };
    auto ld_i = [&](int plane, uint32_t voff) { return SLR_GRAD_BUF_LD ? buf_ld(ri, voff, (uint32_t)plane * hw4) : ip[(size_t)plane * HW + (voff >> 2)]; };
    auto st_o = [&](int plane, float g) {
        if (SLR_GRAD_BUF_ST) buf_st(ro, vst, (uint32_t)plane * hw4, g);
        else if (live_px) op[(size_t)plane * HW + i] = g;
    };
    float gx = 0.0f, gy = 0.0f;
    // staged path: the box is walked as ONE linear index range (dense wave loads across row ends); the values of the NEXT
    // pass are loaded into registers while this pass is gathered from LDS, and written to LDS after the barrier
    constexpr int NS = (GT_BOX + GT_THREADS - 1) / GT_THREADS;
    uint32_t soff[NS];                                             // byte offset of this work-item's k-th box cell inside a plane
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int idx = tid + k * GT_THREADS;
        int qx0 = sx0[0], qy0 = sy0[0], qbw = sbw[0], qb = sbase[0];           // the strip cell idx belongs to
#pragma unroll
        for (int q = 1; q < NSTR; ++q) if (idx >= sbase[q]) { qx0 = sx0[q]; qy0 = sy0[q]; qbw = sbw[q]; qb = sbase[q]; }
        const int rel = idx - qb;
        const int r = (int)(((float)rel + 0.5f) / (float)qbw);     // rel / qbw, exact for rel < 2^22
        soff[k] = (staged && idx < nbox) ? (uint32_t)((qy0 + r) * W + qx0 + (rel - r * qbw)) * 4u : 0u;
    }
    // one pass of U channels: the reference's terms in the reference's order (bit-identical on either path).  FULL: all U channels
    // exist -- every store of the pass is unconditional (counted waits in the loop); the last pass of a channel count that is not a
    // multiple of U branches around the missing ones.
    auto finish = [&](auto full_tag, int ch, const float (&a0)[U], const float (&a1)[U], const float (&a2)[U], const float (&a3)[U], const float (&v)[U]) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!FULL && ch + u >= C) continue;                    // (scalar)
            if (GIN) {
                float g = 0.0f;
                g += k0 ? a0[u] * c.w[0] : 0.0f;
                g += k1 ? a1[u] * c.w[1] : 0.0f;
                g += k2 ? a2[u] * c.w[2] : 0.0f;
                g += k3 ? a3[u] * c.w[3] : 0.0f;
                st_o(ch + u, g);
            }
            if (GFLOW) {                                           // (work-items outside the image: k0..k3 are false, +0.0 everywhere)
                const float t0 = v[u] * a0[u], t1 = v[u] * a1[u], t2 = v[u] * a2[u], t3 = v[u] * a3[u];
                gx += k0 ? t0 * dx[0] : 0.0f; gy += k0 ? t0 * dy[0] : 0.0f;
                gx += k1 ? t1 * dx[1] : 0.0f; gy += k1 ? t1 * dy[1] : 0.0f;
                gx += k2 ? t2 * dx[2] : 0.0f; gy += k2 ? t2 * dy[2] : 0.0f;
                gx += k3 ? t3 * dx[3] : 0.0f; gy += k3 ? t3 * dy[3] : 0.0f;
            }
        }
    };
    if (staged) {                                                  // (workgroup-uniform: each path is a loop of its own)
        const uint32_t b0 = (uint32_t)l0, b1 = (uint32_t)l1, b2 = (uint32_t)l2, b3 = (uint32_t)l3;
        // The staged loop exists once per number of box cells a work-item carries (NSE = ceil(nbox / GT_THREADS), rounded up to 1, 2, 3, 4, 6, NS):
        // a typical Euler block's boxes hold 900-1100 cells, and a loop that always issued NS loads per channel spent three quarters of its
        // vector-memory instructions on cells that do not exist.
        auto run = [&](auto nse_tag) {
            constexpr int NSE = decltype(nse_tag)::value;
            float sv[NSE][U];
            auto issue = [&](int ch) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pl = min(ch + u, C - 1);
#pragma unroll
                    for (int k = 0; k < NSE; ++k) sv[k][u] = ld_g(pl, soff[k]);
                }
            };
            auto pass = [&](auto full_tag, int ch) {
                float a0[U], a1[U], a2[U], a3[U], v[U];
                __syncthreads();                                   // the previous pass has been gathered
#pragma unroll
                for (int k = 0; k < NSE; ++k) {
                    const int idx = tid + k * GT_THREADS;
                    if (idx < nbox)
#pragma unroll
                        for (int u = 0; u < U; ++u) box[u][idx] = sv[k][u];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (GFLOW) v[u] = ld_i(min(ch + u, C - 1), vi);
                __syncthreads();
                issue(ch + U);                                     // in flight under this pass's gathers, sums and stores (past the last
                                                                   // channel: re-reads channel C - 1, unused)
#pragma unroll
                for (int u = 0; u < U; ++u) { a0[u] = box[u][b0]; a1[u] = box[u][b1]; a2[u] = box[u][b2]; a3[u] = box[u][b3]; }
                finish(full_tag, ch, a0, a1, a2, a3, v);
            };
            issue(0);
            int ch = 0;
            for (; ch + U <= C; ch += U) pass(std::true_type{}, ch);
            if (ch < C) pass(std::false_type{}, ch);
        };
        const int need = (nbox + GT_THREADS - 1) / GT_THREADS;    // (workgroup-uniform)
        if (SLR_GRAD_NSE_VARIANTS == 0 || need > 6) run(std::integral_constant<int, NS>{});
        else if (need <= 1) run(std::integral_constant<int, 1>{});
        else if (need == 2) run(std::integral_constant<int, 2>{});
        else if (need == 3) run(std::integral_constant<int, 3>{});
        else if (need == 4) run(std::integral_constant<int, 4>{});
        else run(std::integral_constant<int, 6>{});
    } else if (SLR_GRAD_PAIRS && bent > SLR_GRAD_BENT) {
        // Direct gathers of a block whose rows bend but whose boxes do not fit (a flow that rotates and stretches the block: boxes of 4400-7600
        // cells at Euler t=59; incoherent flows): every lane of a gather instruction sits on a cache line of its own, the block is bound by the
        // line requests its CU's L1 takes (36 of 1920 blocks ran 115-150 us each and were a fifth of the launch).  The two corners of a
        // destination ROW come as one 4-byte-aligned 8-byte load: half the requests.  Same values, same terms: bit-identical.
        const bool kT = k0 | k1, kB = k2 | k3;
        const int safe = min(i, HW - 2);                           // (bent > 2 implies H >= 3)
        const int pT = kT ? min(max(o, 0), HW - 2) : safe, pB = kB ? min(max(o + W, 0), HW - 2) : safe;
        const int dT = o - pT, dB = o + W - pB;                    // -1 (x0 = -1 at the plane's first pixel), 0, +1 (x0 + 1 past its last)
        float2u p0[U], p1[U], q0[U], q1[U];
        float pv[U], qv[U];
        auto load = [&](int ch, float2u (&t)[U], float2u (&b)[U], float (&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pl = min(ch + u, C - 1);
                t[u] = *reinterpret_cast<const float2u *>(gp + (size_t)pl * HW + pT);
                b[u] = *reinterpret_cast<const float2u *>(gp + (size_t)pl * HW + pB);
                if (GFLOW) v[u] = ld_i(pl, vi);
            }
        };
        auto fin = [&](auto full_tag, int ch, const float2u (&t)[U], const float2u (&b)[U], const float (&v)[U]) {
            float a0[U], a1[U], a2[U], a3[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a0[u] = dT == 1 ? t[u].y : t[u].x; a1[u] = dT == -1 ? t[u].x : t[u].y;
                a2[u] = dB == 1 ? b[u].y : b[u].x; a3[u] = dB == -1 ? b[u].x : b[u].y;
            }
            finish(full_tag, ch, a0, a1, a2, a3, v);
        };
        load(0, p0, p1, pv);
        int ch = 0;
        for (; ch + 2 * U <= C; ch += 2 * U) {
            load(ch + U, q0, q1, qv);
            fin(std::true_type{}, ch, p0, p1, pv);
            load(ch + 2 * U, p0, p1, pv);
            fin(std::true_type{}, ch + U, q0, q1, qv);
        }
        if (ch < C) {
            load(ch + U, q0, q1, qv);
            fin(std::false_type{}, ch, p0, p1, pv);
            if (ch + U < C) fin(std::false_type{}, ch + U, q0, q1, qv);
        }
    } else {
        // direct gathers, two register sets: the loads of the NEXT pass are issued before this pass's sums and stores
        const uint32_t v0 = (uint32_t)g0 * 4u, v1 = (uint32_t)g1 * 4u, v2 = (uint32_t)g2 * 4u, v3 = (uint32_t)g3 * 4u;
        float p0[U], p1[U], p2[U], p3[U], pv[U], q0[U], q1[U], q2[U], q3[U], qv[U];
        auto load = [&](int ch, float (&a0)[U], float (&a1)[U], float (&a2)[U], float (&a3)[U], float (&v)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pl = min(ch + u, C - 1);
                a0[u] = ld_g(pl, v0); a1[u] = ld_g(pl, v1); a2[u] = ld_g(pl, v2); a3[u] = ld_g(pl, v3);
                if (GFLOW) v[u] = ld_i(pl, vi);
            }
        };
        load(0, p0, p1, p2, p3, pv);
        int ch = 0;
        for (; ch + 2 * U <= C; ch += 2 * U) {
            load(ch + U, q0, q1, q2, q3, qv);
            finish(std::true_type{}, ch, p0, p1, p2, p3, pv);
            load(ch + 2 * U, p0, p1, p2, p3, pv);                  // (past the last channel: re-reads channel C - 1, unused)
            finish(std::true_type{}, ch + U, q0, q1, q2, q3, qv);
        }
        if (ch < C) {                                              // the last 1 .. 2U - 1 channels
            load(ch + U, q0, q1, q2, q3, qv);
            finish(std::false_type{}, ch, p0, p1, p2, p3, pv);
            if (ch + U < C) finish(std::false_type{}, ch + U, q0, q1, q2, q3, qv);
        }
    }
    if (GFLOW && live_px) {
        // group 0 writes gradFlow itself, group g > 0 its partial sums to gpart[g - 1]
        float *gf = gz > 0 ? gpart + ((size_t)(gz - 1) * gridDim.y + n) * 2 * HW : gflow + (size_t)n * 2 * HW;
        gf[i] = gx;
        gf[HW + i] = gy;
    }
}

// gradFlow = the channel groups' partial sums, added in group order (group 0 wrote gflow itself, group g > 0 gpart[g - 1])
__global__ __launch_bounds__(256) void grad_flow_sum_kernel(const float *__restrict__ gpart, float *__restrict__ gflow, size_t n, int groups) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t = gflow[i];
    for (int g = 1; g < groups; ++g) t += gpart[(size_t)(g - 1) * n + i];
    gflow[i] = t;
}

// out[src] = max(seed[src], max over in-bounds corners of maxwarp[corner])
// (kernel_Inversesplat_updateOutput, softsplat.py:84-155; seed = input.clone(), :606)
__global__ __launch_bounds__(256) void inverse_max_kernel(const float *__restrict__ seed,
                                                          const float *__restrict__ maxwarp,
                                                          const float *__restrict__ flow,
                                                          float *__restrict__ out, int C, int H, int W) {
    const int HW = H * W;
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *f = flow + (size_t)n * 2 * HW;
    const int y = i / W, x = i - y * W;
    const Corners c = make_corners(f[i], f[HW + i], x, y);
    const bool k0 = c.ok & in_image(c.x0, c.y0, H, W), k1 = c.ok & in_image(c.x0 + 1, c.y0, H, W);
    const bool k2 = c.ok & in_image(c.x0, c.y0 + 1, H, W), k3 = c.ok & in_image(c.x0 + 1, c.y0 + 1, H, W);
    const int o = c.y0 * W + c.x0;
    const float *sp = seed + (size_t)n * C * HW;
    const float *mp = maxwarp + (size_t)n * C * HW;
    float *op = out + (size_t)n * C * HW;
    for (int ch = 0; ch < C; ++ch, sp += HW, mp += HW, op += HW) {
        float m = sp[i];
        if (k0) m = fmaxf(mp[o], m);
        if (k1) m = fmaxf(mp[o + 1], m);
        if (k2) m = fmaxf(mp[o + W], m);
        if (k3) m = fmaxf(mp[o + W + 1], m);
        op[i] = m;
    }
}

}  // namespace slr

using namespace slr;

// Channel groups of the tiled kernel (its 64 KiB box: two workgroups per CU = 512 slots): on grids smaller than the chip as many groups as
// keep the launch within ~2 rounds of the slots, on larger ones two; each a multiple of the kernel's 4 channels per pass.
static int grad_groups(int N, int C, int H, int W) {
    const long long wgs = (long long)N * ((W + TILE_W - 1) / TILE_W) * ((H + TILE_H - 1) / TILE_H);
    int g = (int)(SLR_GRAD_SLOTS / (wgs > 0 ? wgs : 1));
    g = g > SLR_GRAD_GROUPS_MAX ? SLR_GRAD_GROUPS_MAX : g;
    g = g < SLR_GRAD_GROUPS_MIN ? SLR_GRAD_GROUPS_MIN : g;
    const int byc = C / 8;
    g = g > byc ? byc : g;
    return g < 1 ? 1 : g;
}

SLR_EXPORT size_t slr_softsplat_backward_ws_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if ((long long)C * H * W * 4 >= (1LL << 31)) return 0;      // (plane stacks of 2 GiB and more per sample: grad_kernel, no groups)
    const int g = grad_groups(N, C, H, W);
    return g > 1 ? (size_t)(g - 1) * N * 2 * H * W * 4 : 0;
}

SLR_EXPORT int slr_softsplat_backward_ws(const float *in, const float *flow, const float *grad_out, float *grad_in,
                                         float *grad_flow, int N, int C, int H, int W, void *ws, size_t ws_bytes, void *stream) {
    SLR_CHECK_ARG(flow && grad_out, "null pointer");
    SLR_CHECK_ARG(!grad_flow || in, "input required for grad_flow");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (long long)N * H * W < (1LL << 29), "sizes");
    hipStream_t st = (hipStream_t)stream;
    // the tiled kernel addresses one sample's C planes through a 32-bit buffer descriptor
    const bool tiled = (long long)C * H * W * 4 < (1LL << 31);
    if (tiled) {
        const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
        // channel groups: only with scratch for the partial gradFlow sums (or when gradFlow is not asked for)
        int groups = grad_groups(N, C, H, W);
        if (grad_flow && groups > 1 && (!ws || ws_bytes < (size_t)(groups - 1) * N * 2 * H * W * 4)) groups = 1;
        const int cper = groups > 1 ? ((C + groups - 1) / groups + SLR_GRAD_TU - 1) / SLR_GRAD_TU * SLR_GRAD_TU : C;
        groups = (C + cper - 1) / cper;
        float *gpart = (float *)ws;
        dim3 grid(tiles_x * tiles_y, N, groups);
        if (grad_in && grad_flow) hipLaunchKernelGGL((grad_tile_kernel<true, true>), grid, dim3(GT_THREADS), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W, tiles_x, cper, gpart);
        else if (grad_in) hipLaunchKernelGGL((grad_tile_kernel<true, false>), grid, dim3(GT_THREADS), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W, tiles_x, cper, gpart);
        else if (grad_flow) hipLaunchKernelGGL((grad_tile_kernel<false, true>), grid, dim3(GT_THREADS), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W, tiles_x, cper, gpart);
        if (grad_flow && groups > 1) {
            const size_t n = (size_t)N * 2 * H * W;
            hipLaunchKernelGGL(grad_flow_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)gpart, grad_flow, n, groups);
        }
    } else {
        dim3 grid((H * W + 255) / 256, N);
        if (grad_in && grad_flow) hipLaunchKernelGGL((grad_kernel<true, true>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
        else if (grad_in) hipLaunchKernelGGL((grad_kernel<true, false>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
        else if (grad_flow) hipLaunchKernelGGL((grad_kernel<false, true>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
    }
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_softsplat_backward(const float *in, const float *flow, const float *grad_out, float *grad_in,
                                      float *grad_flow, int N, int C, int H, int W, void *stream) {
    return slr_softsplat_backward_ws(in, flow, grad_out, grad_in, grad_flow, N, C, H, W, nullptr, 0, stream);
}

SLR_EXPORT int slr_max_warp_norm(const float *in, const float *flow, float *scratch, float *out, int N, int C,
                                 int H, int W, void *ws, size_t ws_bytes, int prebinned, void *stream) {
    SLR_CHECK_ARG(in && flow && scratch && out, "null pointer");
    // max-splat seeded with -1000 (softsplat.py:590) ...
    if (int e = slr_maxsplat_forward(in, flow, scratch, -1000.0f, N, C, H, W, ws, ws_bytes, prebinned, stream))
        return e;
    // ... then gather back per source pixel (softsplat.py:606-618)
    dim3 grid((H * W + 255) / 256, N);
    hipLaunchKernelGGL(inverse_max_kernel, grid, dim3(256), 0, (hipStream_t)stream, in,
                       (const float *)scratch, flow, out, C, H, W);
    SLR_CHECK_LAUNCH();
    return 0;
}
