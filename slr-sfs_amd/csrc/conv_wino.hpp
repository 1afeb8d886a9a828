// conv_wino.hpp -- the fp32 rung's 3x3 convolution as Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32 (gfx950); included by conv.hip.
//
// The fp32 rung (SLR_CONV_F32: fp32 operands, fp32 products, fp32 accumulation -- the arithmetic class of the reference's decoder,
// models/layers/partialconv2d.py:61-74) runs at 0.85 of the fp32 matrix peak (157 TFLOP/s) as a direct implicit GEMM: it can only get
// faster by doing fewer multiplications.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input patch with 16 multiplications per
// (input channel, output channel) instead of 36:  Y = A^T [ (G g G^T) .* (B^T d B) ] A, summed over the input channels BEFORE the
// output transform -- i.e. 16 independent GEMMs (one per position xi of the 4x4 transformed tile) of [couts x cins] x [cins x tiles].
//
// One workgroup (256 work-items = 4 waves, TWO workgroups per CU: 80 KB of LDS each, 128 accumulator registers per wave) = an 8 x 16
// output block (32 tiles of 2x2) x 64 output channels:
//   * staging: the (8+2) x (16+2) halo block of 16 input channels through the same prologue as the direct kernel
//     (relu(x*scale - shift)*mask, zero padding) as fp32 rows in LDS (`raw`, single buffer): the global loads of chunk c + 3 are issued
//     behind the weight loads of chunk c's last pair, chunk c + 2 is stored piece by piece in slots 15-22 of chunk c (branch-free);
//   * input transform: every work-item turns 2 patches (2 channels of one tile) into V[xi][cin][tile] (32 additions per patch),
//     double-buffered in LDS -- the transform of chunk c + 1 runs in 7 slices between the MFMAs of chunk c (slots 0-6 and 8-14);
//   * wave (cot, xh) owns 32 output channels x the 32 tiles x 8 of the 16 positions: 8 accumulator tiles of 32x32; per chunk and
//     position 8 MFMAs (K = 2 input channels each) whose B operand is ONE ds_read_b32 of V and whose A operand (the transformed
//     weights U = G g G^T, prepared by slr_conv3x3_wino_weights in fragment order) comes from L2, a pair of positions ahead;
//   * after the last chunk the two halves of the positions meet through LDS; output transform in registers, then the direct kernel's
//     epilogues (plain + bias + residual, or the partial-convolution one with its mask box sum, next-layer BN, update mask) on the
//     2x2 pixels, per-channel constants from an LDS table, the residual requested before the exchange.
// 2.25x fewer MFMAs than the direct kernel per output; the transforms are additions only.  Accuracy: per layer 2 - 4x the direct fp32
// kernel's error against fp64 (tests/test_gpu_conv_f32.py measures both); what that does to whole frames of an ill-conditioned network
// is in DESIGN.md 3.4 -- the reason this is a rung of its own (convs="fp32-winograd"), not the strict fp32 rung.
#pragma once
#include <type_traits>

namespace slr {

constexpr int WN_BW = 16, WN_BH = 8;                 // output block of a workgroup: 8 rows x 16 columns = 32 tiles of 2x2 (4 tile rows x 8 tile columns)
constexpr int WN_HW = WN_BW + 2, WN_HH = WN_BH + 2;  // its input halo block
constexpr int WN_NPX = WN_HW * WN_HH;                // 180 halo pixels
constexpr int WN_RAWSTR = 184;                       // floats per channel row of the staged halo block (180 used)
constexpr int WN_TILES = 32;
constexpr int WN_MAXCIN = 256;                       // prologue scale / shift table (wider layers take the direct kernel)
constexpr size_t WN_OFF_RAW = 0;
constexpr size_t WN_OFF_V = WN_OFF_RAW + (size_t)16 * WN_RAWSTR * 4;                       // [2][16 xi][16 cin][32 tiles]
constexpr size_t WN_OFF_MPL = WN_OFF_V + (size_t)2 * 16 * 16 * WN_TILES * 4;
constexpr size_t WN_OFF_PSS = (WN_OFF_MPL + (size_t)WN_NPX * 4 + 15) & ~(size_t)15;
constexpr size_t WN_OFF_EPI = WN_OFF_PSS + (size_t)2 * WN_MAXCIN * 4;                       // bias | next scale | next shift of the 64 output channels
constexpr size_t WN_LDS_BYTES = WN_OFF_EPI + (size_t)3 * 64 * 4;
static_assert(2 * WN_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__host__ __device__ inline int wino_cout_pad(int Cout) { return (Cout + 63) / 64 * 64; }

// w [Cout,Cin,3,3] -> U = G g G^T per (co, ci), fp32, in fragment order [co tile 32][chunk of 16 ci][xi 16][lane 64][k pair 8]:
// lane = (co & 31) + 32 * (ci & 1), k = (ci & 15) >> 1 (the A operand of v_mfma_f32_32x32x2_f32: lanes 0-31 hold input channel 2k,
// lanes 32-63 channel 2k + 1).  Computed in double, rounded once.
__global__ __launch_bounds__(256) void conv_wino_weights_kernel(const float *__restrict__ w, float *__restrict__ wf, int Cout, int Cin,
                                                                int CoutP, int CinP) {
    const int total = CoutP * CinP, nchunk = CinP >> 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % CinP, co = i / CinP;
        double g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) g[r][c] = (co < Cout && ci < Cin) ? (double)w[((size_t)co * Cin + ci) * 9 + r * 3 + c] : 0.0;
        double t[4][3];                                  // G g
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t[0][c] = g[0][c];
            t[1][c] = 0.5 * (g[0][c] + g[1][c] + g[2][c]);
            t[2][c] = 0.5 * (g[0][c] - g[1][c] + g[2][c]);
            t[3][c] = g[2][c];
        }
        const int cl = ci & 15;
        const size_t frag0 = ((size_t)(co >> 5) * nchunk + (ci >> 4)) * 16;
        const int within = ((co & 31) + 32 * (cl & 1)) * 8 + (cl >> 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {                    // (G g) G^T
            const double u[4] = {t[r][0], 0.5 * (t[r][0] + t[r][1] + t[r][2]), 0.5 * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
            for (int c = 0; c < 4; ++c) wf[(frag0 + r * 4 + c) * 512 + within] = (float)u[c];
        }
    }
}

#define WN_BDEPTH 1      // slots the B operand reads run ahead of their MFMAs (2, 3: no change, 1269 - 1275 us at 128 -> 128)
#define WN_SLICE 8       // blocks per dispatch slice: the workgroups of a block's output-channel groups are WN_SLICE apart in dispatch order (32 .. 256: no change)
constexpr int WN_THREADS = 256;                      // 4 waves; two INDEPENDENT workgroups per CU (one wave of each per SIMD): while one stands at a
                                                     // barrier or stores its staged block the other keeps the matrix pipe busy (8 coupled waves: 1604 us)

#ifdef SLR_TRACE
#define WN_STAMP(slot) do { if (a.trace && threadIdx.x == 0) a.trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + (slot)] = clock64(); } while (0)
#else
#define WN_STAMP(slot) do { } while (0)
#endif

template <bool PRE, bool INB8>
__global__ __launch_bounds__(WN_THREADS, 2) void conv3x3_wino_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wn_smem[];
    float (*raw)[WN_RAWSTR] = reinterpret_cast<float (*)[WN_RAWSTR]>(wn_smem + WN_OFF_RAW);                       // [16 cin][halo pixel]
    float (*V)[16][16][WN_TILES] = reinterpret_cast<float (*)[16][16][WN_TILES]>(wn_smem + WN_OFF_V);            // [buf][xi][cin][tile]
    float *mpl = reinterpret_cast<float *>(wn_smem + WN_OFF_MPL);                                                // mask plane over the halo block
    float *pss = reinterpret_cast<float *>(wn_smem + WN_OFF_PSS);                                                // prologue scale | shift
    float *epi = reinterpret_cast<float *>(wn_smem + WN_OFF_EPI);                                                // epilogue bias | scale | shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave: output channels 32 * cot .. + 31 of the workgroup's 64, all 32 tiles, positions 8 * xh .. + 7
    const int cot = wave & 1, xh = wave >> 1;
    constexpr int tb = 0;
    // grid.x = output blocks x groups of 64 output channels, the groups of one block 8 workgroups apart: dispatched together and (block b
    // runs on XCD b % 8, observed) on the SAME XCD -- the second group's staging loads find the block's input in that L2 instead of in HBM
    // (their latency also sits in front of every weight load issued behind them: memory returns in order)
    const int ngrp = a.wino_groups;                      // groups of 64 output channels
    const int bsl = blockIdx.x / (WN_SLICE * ngrp), brem = blockIdx.x - bsl * WN_SLICE * ngrp;
    const int blk = bsl * WN_SLICE + (brem % WN_SLICE), cgrp = brem / WN_SLICE;
    if (blk >= a.tiles_x * ((a.H + WN_BH - 1) / WN_BH)) return;       // (the last slice of 8 blocks may be short; whole workgroups)
    WN_STAMP(0);
    const int tx = blk % a.tiles_x, ty = blk / a.tiles_x;
    const int x0 = tx * WN_BW, y0 = ty * WN_BH;
    const int n = blockIdx.z;
    const int cotile = cgrp * 2 + cot;                   // 32-channel tile of the weight buffer
    const int HW = a.H * a.W;
    const int nchunk = a.nchunk;
    const int cmax = a.Cin - 1;
    const float *inb = a.in + (size_t)n * a.Cin * HW;
    const int pre = a.pre;

    // ---- staging of a chunk: 360 items = 180 halo pixels x 2 groups of 8 channels; item A = tid (all work-items), item B = 256 + tid
    // (work-items < 104).  Same prologue as the direct kernel (conv.hip: stage_value).
    const bool liveB = tid < 2 * WN_NPX - WN_THREADS;
    const int gA = tid >= WN_NPX ? 1 : 0, pA = tid - WN_NPX * gA;      // group / halo pixel of item A
    const int pB = WN_THREADS - WN_NPX + tid;                          // item B: group 1, pixels 76 .. 179
    bool okA, okB;
    int offA, offB;
    {
        const int pr = pA / WN_HW, pc = pA - pr * WN_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        okA = (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        offA = okA ? gy * a.W + gx : 0;
    }
    {
        const int pq = liveB ? pB : 0, pr = pq / WN_HW, pc = pq - pr * WN_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        okB = liveB & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        offB = okB ? gy * a.W + gx : 0;
    }
    const float mvA = (a.mask && okA) ? a.mask[(size_t)n * HW + offA] : 0.0f;
    const float mvB = (a.mask && okB) ? a.mask[(size_t)n * HW + offB] : 0.0f;
    const float mA = okA ? (pre == PRE_BN_MASK ? mvA : 1.0f) : 0.0f;
    const float mB = okB ? (pre == PRE_BN_MASK ? mvB : 1.0f) : 0.0f;
    float cntA = 0.0f, cntB = 0.0f;                    // derived mask: non-zero inputs per staging item, over all chunks
    const bool nonzero_mask = pre == PRE_BN_NONZERO;
    const int c8max = (a.Cin >> 3) - 1;
    auto load_item = [&](bool isB, int c, float (&st)[8]) {
        const int g = isB ? 1 : gA, off = isB ? offB : offA;
        if (INB8) {
            const int grp = min(c * 2 + g, c8max);
            const float4 *q = reinterpret_cast<const float4 *>(inb) + ((size_t)grp * HW + (unsigned)off) * 2;
            const float4 u = q[0], v = q[1];
            st[0] = u.x; st[1] = u.y; st[2] = u.z; st[3] = u.w;
            st[4] = v.x; st[5] = v.y; st[6] = v.z; st[7] = v.w;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) st[j] = inb[(size_t)min(c * 16 + g * 8 + j, cmax) * HW + (unsigned)off];
    };
    // (no branch around a piece: dead work-items of item B write the row's 4 padding floats, pieces past the last chunk re-stage the last
    // chunk from stale registers into the buffer nobody reads -- 16 branches per chunk between the MFMAs cost 0.5 us of its 4.6)
    const int pBst = liveB ? pB : WN_NPX + (tid & 3);
    // (have: sc / sh are the prologue's scale / shift of the channel, fetched ahead of time by the caller; otherwise read here)
    auto store_piece = [&](bool isB, int c, const float (&st)[8], int j, float counted = 1.0f, bool have = false, float sc = 0.0f, float sh = 0.0f) {       // channel j of the item's 8
        const int g = isB ? 1 : gA, px = isB ? pBst : pA;
        const int cb = c * 16 + g * 8;
        const float mk0 = isB ? mB : mA;
        const float x = st[j];
        float v;
        if (PRE) {
            if (!have) { sc = pss[cb + j]; sh = pss[WN_MAXCIN + cb + j]; }
            float mk = mk0;
            if (!INB8) {                                 // the derived mask (x != 0) comes with NCHW input only (conv.hip: conv_check_layout):
                mk = (nonzero_mask & (x == 0.0f)) ? 0.0f : mk0;         // the channel-blocked instantiation carries neither the test nor the count
                const float count = (cb + j <= cmax) ? mk * counted : 0.0f;
                if (isB) cntB += count; else cntA += count;
            }
            v = fmaxf(x * sc - sh, 0.0f) * mk;
        } else {
            v = (cb + j <= cmax) ? x * mk0 : 0.0f;       // (padded channels meet zero weights; keep them finite and zero)
        }
        raw[g * 8 + j][px] = v;
    };
    auto store_item = [&](bool isB, int c, const float (&st)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) store_piece(isB, c, st, j);
    };

    // ---- input transform: work-item -> tile tid & 31, channels 2 * (tid >> 5), + 1 of the chunk
    const int tt = tid & 31, tcg = tid >> 5;
    const int tpy = (tt >> 3) * 2, tpx = (tt & 7) * 2;   // the patch's first halo row / column
    // A patch's transform in 7 slices, issued between the MFMA groups (a slice of <= 16 instructions issues in the shadow of the matrix
    // pipe): 0 patch loads, 1-2 B^T d, 3-6 one output row each.
    struct PatchRegs { float d[4][4], t[4][4]; };
    auto transform_slice = [&](int slice, int vb, int k, PatchRegs &pr) {
        const int ch = tcg * 2 + k;
        if (slice == 0) {
            const float *rp = &raw[ch][tpy * WN_HW + tpx];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 lo = *reinterpret_cast<const float2 *>(rp + i * WN_HW), hi = *reinterpret_cast<const float2 *>(rp + i * WN_HW + 2);
                pr.d[i][0] = lo.x; pr.d[i][1] = lo.y; pr.d[i][2] = hi.x; pr.d[i][3] = hi.y;
            }
        } else if (slice == 1 || slice == 2) {           // B^T d, two columns per slice
#pragma unroll
            for (int j = 2 * (slice - 1); j < 2 * slice; ++j) {
                pr.t[0][j] = pr.d[0][j] - pr.d[2][j];
                pr.t[1][j] = pr.d[1][j] + pr.d[2][j];
                pr.t[2][j] = pr.d[2][j] - pr.d[1][j];
                pr.t[3][j] = pr.d[1][j] - pr.d[3][j];
            }
        } else if (slice <= 6) {                         // (B^T d) B, one row per slice
            const int i = slice - 3;
            V[vb][i * 4 + 0][ch][tt] = pr.t[i][0] - pr.t[i][2];
            V[vb][i * 4 + 1][ch][tt] = pr.t[i][1] + pr.t[i][2];
            V[vb][i * 4 + 2][ch][tt] = pr.t[i][2] - pr.t[i][1];
            V[vb][i * 4 + 3][ch][tt] = pr.t[i][1] - pr.t[i][3];
        }
    };

    f16v acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;

    const int bcol = lane & 31, bgrp = lane >> 5;
    typedef float f8v __attribute__((ext_vector_type(8)));
    const float4 *wbase = reinterpret_cast<const float4 *>(a.w) + ((size_t)cotile * nchunk * 16 + 8 * xh) * 128;       // 128 float4 per (chunk, xi) fragment
    auto load_a = [&](int g /* chunk * 16 + local xi */) -> f8v {
        const float4 *q = wbase + (size_t)g * 128 + 2u * (unsigned)lane;
        const float4 u = q[0], v = q[1];
        f8v r;
        r[0] = u.x; r[1] = u.y; r[2] = u.z; r[3] = u.w; r[4] = v.x; r[5] = v.y; r[6] = v.z; r[7] = v.w;
        return r;
    };
    // A wave's 8 positions go two at a time ("pair": 8 k-pairs x 2 accumulators), the pair's two weight fragments loaded a whole pair
    // ahead of their use: the loads are ISSUED at the top of the previous pair (scheduling barrier) -- sunk to their use by the
    // compiler, every pair waited for an L2 round trip (2255 -> 1749 us at 128 -> 128, 768x1280).  Local pair p of chunk c: fragments
    // c * 16 + 2p, + 1 (wbase already points at this wave's half of the positions).
    const int nlast = (nchunk - 1) * 16 + 7;
    auto frag = [&](int c, int j) { return min(c * 16 + j, nlast); };
    f8v aq[2], an[2];
    aq[0] = load_a(frag(0, 0)); aq[1] = load_a(frag(0, 1));

    // ---- prologue.  Everything the first chunks need from global memory is requested before anything is waited for: chunks 0 and 1, the
    // first weight fragments, then the per-channel tables (one memory round trip in front of the first MFMA instead of three).
    float s0A[8], s0B[8], sA[8], sB[8];
    load_item(false, 0, s0A);
    load_item(true, 0, s0B);
    if (nchunk > 1) {
        load_item(false, 1, sA);
        load_item(true, 1, sB);
    }
    if (a.mask && tid < WN_NPX) mpl[tid] = mvA;        // (items A of group 0 cover every halo pixel)
    if (PRE) {
        for (int i = tid; i < nchunk * 16; i += WN_THREADS) {       // padded channels: scale = shift = 0 -> 0
            pss[i] = i < a.Cin ? a.pre_scale[i] : 0.0f;
            pss[WN_MAXCIN + i] = i < a.Cin ? a.pre_shift[i] : 0.0f;
        }
    }
    if (tid < 64) {                                    // the epilogue's per-channel constants: read from LDS there, no global round trip between
        const int cc = min(cgrp * 64 + tid, a.Cout - 1);          // the exchange and the stores
        epi[tid] = a.bias ? a.bias[cc] : 0.0f;
        epi[64 + tid] = a.next_scale ? a.next_scale[cc] : 1.0f;
        epi[128 + tid] = a.next_scale ? a.next_shift[cc] : 0.0f;
    }
    __syncthreads();                                   // tables
    store_item(false, 0, s0A);
    store_item(true, 0, s0B);
    __syncthreads();
    if (nchunk > 2) {                                  // chunk 2 flies under the first transform
        load_item(false, 2, s0A);
        load_item(true, 2, s0B);
    }
    {
        PatchRegs pr;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int sl = 0; sl < 7; ++sl) transform_slice(sl, 0, k, pr);
    }
    __syncthreads();                                   // V[0] complete, raw read
    if (nchunk > 1) {                                  // raw <- chunk 1: the loop's invariant (raw holds chunk c + 1 when chunk c begins)
        store_item(false, 1, sA);
        store_item(true, 1, sB);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sA[j] = s0A[j]; sB[j] = s0B[j]; }        // (the loop's staging registers hold chunk c + 2)
    __syncthreads();
    WN_STAMP(1);

    // A chunk's 32 slots (pair p, k-pair kp: slot 8p + kp), two MFMAs each.  In their shadow:
    //   slots 0-6   patch 0 of chunk c + 1: raw -> V[other buffer]      slots 8-14  patch 1 (raw is read in slots 0 and 8 only)
    //   slot 9      barrier: raw is free
    //   slots 15-22 raw <- chunk c + 2 (prologue applied), one of the 8 channels of both staging items per slot
    //   slot 24     the loads of chunk c + 3 are issued, BEHIND pair 3's weight loads: memory returns in order, so they have until the end of
    //               the next chunk's pair 0 (two pairs), and nothing waits for them before
    // No phase in which the matrix pipe waits for the staging (stores between two barriers at the top of the chunk: 1543 us -> see DESIGN).
    auto read_b = [&](int vb, int slot, int i) -> float {       // B operand of slot 8p + kp, accumulator i of its pair
        return V[vb][8 * xh + 2 * (slot >> 3) + i][2 * (slot & 7) + bgrp][tb * 32 + bcol];
    };
    float bq[WN_BDEPTH][2];
    for (int c = 0; c < nchunk; ++c) {
        const int vb = c & 1;
        const bool stage_ld = c + 3 < nchunk;          // (uniform)
        const int cst = min(c + 2, nchunk - 1);
        const float fst = c + 2 < nchunk ? 1.0f : 0.0f;                      // (the derived mask counts a chunk once)
        float2 pq[2][4];                               // prologue constants in flight: [set][scale A, shift A, scale B, shift B] of two channels
        PatchRegs pr;                                  // (the last chunk transforms stale rows into the buffer nobody reads: no branch in the pairs)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // next pair's fragments: p + 1 of this chunk, or pair 0 of the next chunk
            an[0] = load_a(p < 3 ? frag(c, 2 * p + 2) : frag(c + 1, 0));
            an[1] = load_a(p < 3 ? frag(c, 2 * p + 3) : frag(c + 1, 1));
            if (p == 3 && stage_ld) {
                load_item(false, c + 3, sA);
                load_item(true, c + 3, sB);
            }
            if (p == 0) {                              // (the B operands run WN_BDEPTH slots ahead, across the pairs of a chunk)
#pragma unroll
                for (int d = 0; d < WN_BDEPTH; ++d)
#pragma unroll
                    for (int i = 0; i < 2; ++i) bq[d][i] = read_b(vb, d, i);
            }
            __builtin_amdgcn_sched_barrier(0);         // the next pair's weights are in flight from HERE
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) {
                const int slot = p * 8 + kp;
                if (slot == 9) __syncthreads();
                float b[2] = {bq[0][0], bq[0][1]};
#pragma unroll
                for (int d = 0; d + 1 < WN_BDEPTH; ++d) { bq[d][0] = bq[d + 1][0]; bq[d][1] = bq[d + 1][1]; }
                if (slot + WN_BDEPTH < 32) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) bq[WN_BDEPTH - 1][i] = read_b(vb, slot + WN_BDEPTH, i);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[2 * p + i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[i][kp], b[i], acc[2 * p + i], 0, 0, 0);
                if (slot < 15 && (slot & 7) < 7) transform_slice(slot & 7, vb ^ 1, slot >> 3, pr);
                if (PRE && (slot == 13 || slot == 15 || slot == 17 || slot == 19)) {       // scale / shift of the next two pieces' channels, two slots
                    const int jj = slot - 13, set = (jj >> 1) & 1;                          // ahead of their use (read in the piece itself, every piece
                    const int ca = cst * 16 + gA * 8 + jj, cbb = cst * 16 + 8 + jj;          // waited an LDS round trip: 1395 -> 1345 us)
                    pq[set][0] = *reinterpret_cast<const float2 *>(&pss[ca]);
                    pq[set][1] = *reinterpret_cast<const float2 *>(&pss[WN_MAXCIN + ca]);
                    pq[set][2] = *reinterpret_cast<const float2 *>(&pss[cbb]);
                    pq[set][3] = *reinterpret_cast<const float2 *>(&pss[WN_MAXCIN + cbb]);
                }
                if (slot >= 15 && slot < 23) {
                    const int j = slot - 15, set = (j >> 1) & 1;
                    const bool odd = j & 1;
                    store_piece(false, cst, sA, j, fst, PRE, odd ? pq[set][0].y : pq[set][0].x, odd ? pq[set][1].y : pq[set][1].x);
                    store_piece(true, cst, sB, j, fst, PRE, odd ? pq[set][2].y : pq[set][2].x, odd ? pq[set][3].y : pq[set][3].x);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            aq[0] = an[0]; aq[1] = an[1];
        }
        __syncthreads();
        if (c < 8) WN_STAMP(2 + c);
    }

    if (pre == PRE_BN_NONZERO) {                       // mask plane = channel sum of (x != 0) over both channel groups of a pixel
        float *c0 = &raw[0][0], *c1 = c0 + WN_NPX;     // (raw is free: the last transform has been read)
        (gA ? c1 : c0)[pA] = cntA;
        if (liveB) c1[pB] = cntB;
        __syncthreads();
        if (tid < WN_NPX) mpl[tid] = c0[tid] + c1[tid];
        __syncthreads();
    }

    // ---- epilogue.  This lane: tile tl of the block, i.e. output pixels (2 * (tl >> 3) + dy, 2 * (tl & 7) + dx); accumulator register r:
    // output channel 32 * cotile + (r & 3) + 8 * (r >> 2) + 4 * bgrp.  Everything that needs global memory (the residual) is requested
    // BEFORE the exchange, the per-channel constants come from LDS: between the exchange and the stores there is arithmetic only
    // (bias / residual loads inside the channel loop: two L2 round trips per workgroup, 6.6 us of its 46 -> DESIGN).
    const bool partial = a.partial != 0, has_res = a.residual != nullptr, has_next = a.next_scale != nullptr;
    const int tl = tb * 32 + bcol;
    const int py = (tl >> 3) * 2, px = (tl & 7) * 2;
    const int cout1 = a.Cout - 1;
    const float mscale = a.mask_scale, winsize = a.winsize;
    bool ok[4];
    size_t pix[4];
    float um[4], ratio[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int dy = q >> 1, dx = q & 1;
        const int oy = y0 + py + dy, ox = x0 + px + dx;
        ok[q] = (oy < a.H) & (ox < a.W);
        pix[q] = ok[q] ? (size_t)oy * a.W + ox : 0;
        um[q] = 1.0f; ratio[q] = 1.0f;
        if (partial) {
            float box = 0.0f;                            // conv(mask, ones): 3x3 box sum, zero padded (partialconv2d.py:61)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) box += mpl[(py + dy + i) * WN_HW + px + dx + j];
            const float u = box * mscale;
            um[q] = fminf(fmaxf(u, 0.0f), 1.0f);
            ratio[q] = (1.0f / (u + 1e-8f)) * winsize * um[q];
            if (ok[q] && a.um_out && cgrp == 0 && cot == 0 && xh == 0 && bgrp == 0) a.um_out[(size_t)n * HW + pix[q]] = um[q];
        }
    }
    // residual (the last operation of both epilogues), in its own layout: [channel group gg][channel of the group][pixel q]
    float rv[2][4][4];
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        const int g4 = 2 * xh + gg, c8 = cotile * 4 + g4;           // 8-channel group of the blocked layouts
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) rv[gg][rr][q] = 0.0f;
        if (has_res) {
            if (a.res_b8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool live = ok[q] && c8 * 8 + 4 * bgrp < a.Cout;
                    const size_t bidx = (((size_t)n * (a.Cout >> 3) + (live ? c8 : 0)) * HW + pix[q]) * 8 + 4 * bgrp;
                    const float4 v = *reinterpret_cast<const float4 *>(&a.residual[bidx]);
                    rv[gg][0][q] = v.x; rv[gg][1][q] = v.y; rv[gg][2][q] = v.z; rv[gg][3][q] = v.w;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rv[gg][rr][q] = a.residual[((size_t)n * a.Cout + min(cotile * 32 + rr + 8 * g4 + 4 * bgrp, cout1)) * HW + pix[q]];
            }
        }
    }

    // ---- the two halves of the positions meet: a wave finishes the output channels (accumulator rows) 8 * xh .. + 7 of its tile and hands
    // the other 8 rows of its 8 positions to its partner (same cot) through the V area (16 KiB per wave, all of V)
    float *xch = reinterpret_cast<float *>(wn_smem + WN_OFF_V);
    // (xh is a run-time value: accumulator rows indexed with it go through the GPR index mode, one element per s_set_gpr_idx_on / v_mov /
    // _off -- 192 of them, 6 us per workgroup.  Both halves are instantiated with xh as a constant and the wave branches once.)
    auto hand_over = [&](auto half) {
        constexpr int XH = decltype(half)::value;
        float *mine = xch + (size_t)wave * 64 * 64;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) mine[(j * 8 + rr) * 64 + lane] = acc[j][8 * (1 - XH) + rr];
        if (XH) asm volatile("; hand_over, upper half"); else asm volatile("; hand_over, lower half");       // (keeps the two instantiations apart: merged, the
    };                                                                                                     // row index is a select again)
    if (xh == 0) hand_over(std::integral_constant<int, 0>()); else hand_over(std::integral_constant<int, 1>());
    __syncthreads();
    WN_STAMP(10);
    const float *theirs = xch + (size_t)(wave ^ 2) * 64 * 64;

    const bool pair_ok = !a.out_b8 && (a.W % 2 == 0) && !(((uintptr_t)a.out | (uintptr_t)a.residual) & 7);     // 8-byte row pairs
    auto finish = [&](auto half) {
    constexpr int XH = decltype(half)::value;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {                   // four channels at a time (one 16-byte group of the channel-blocked layout)
        const int g4 = 2 * XH + gg;
        float o[4][4];                                 // [channel of the group][pixel q]
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = g4 * 4 + rr, rl = gg * 4 + rr;           // accumulator row; its index among this wave's 8 rows
            float m[16];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float own = acc[j][r], other = theirs[(j * 8 + rl) * 64 + lane];
                m[8 * XH + j] = own;
                m[8 * (1 - XH) + j] = other;
            }
            float t0[4], t1[4];                          // A^T m
#pragma unroll
            for (int j = 0; j < 4; ++j) { t0[j] = m[j] + m[4 + j] + m[8 + j]; t1[j] = m[4 + j] - m[8 + j] - m[12 + j]; }
            o[rr][0] = t0[0] + t0[1] + t0[2]; o[rr][1] = t0[1] - t0[2] - t0[3];
            o[rr][2] = t1[0] + t1[1] + t1[2]; o[rr][3] = t1[1] - t1[2] - t1[3];
        }
        int co[4];
        float eb[4], esc[4], esh[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int cl = cot * 32 + rr + 8 * g4 + 4 * bgrp;      // channel among the workgroup's 64
            co[rr] = cgrp * 64 + cl;
            eb[rr] = epi[cl]; esc[rr] = epi[64 + cl]; esh[rr] = epi[128 + cl];
        }
        const int c8 = cotile * 4 + g4;                // 8-channel group of the blocked layouts
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = o[rr][q];
                if (partial) {
                    v = (v * ratio[q] + eb[rr]) * um[q];                                         // partialconv2d.py:72-74
                    v += rv[gg][rr][q];                                                          // blocks.py:248
                    if (has_next) v = fmaxf(v * esc[rr] - esh[rr], 0.0f) * um[q];                // blocks.py:233-236
                } else {
                    v += eb[rr];
                    v += rv[gg][rr][q];
                }
                o[rr][q] = v;
            }
        if (gg == 0) WN_STAMP(14); else WN_STAMP(15);
        if (a.out_b8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool live = ok[q] && c8 * 8 + 4 * bgrp < a.Cout;
                const size_t bidx = (((size_t)n * (a.Cout >> 3) + (live ? c8 : 0)) * HW + pix[q]) * 8 + 4 * bgrp;
                if (live) *reinterpret_cast<float4 *>(&a.out[bidx]) = make_float4(o[0][q], o[1][q], o[2][q], o[3][q]);
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (co[rr] > cout1) continue;
                float *op = a.out + ((size_t)n * a.Cout + co[rr]) * HW;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    if (pair_ok && ok[2 * dy] && ok[2 * dy + 1]) *reinterpret_cast<float2 *>(op + pix[2 * dy]) = make_float2(o[rr][2 * dy], o[rr][2 * dy + 1]);
                    else {
                        if (ok[2 * dy]) op[pix[2 * dy]] = o[rr][2 * dy];
                        if (ok[2 * dy + 1]) op[pix[2 * dy + 1]] = o[rr][2 * dy + 1];
                    }
                }
            }
        }
    }
    };
    if (xh == 0) finish(std::integral_constant<int, 0>()); else finish(std::integral_constant<int, 1>());
    WN_STAMP(11);
#ifdef SLR_TRACE
    if (a.trace && threadIdx.x == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); a.trace[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 12] = hw | ((long long)xcc << 32); }
#endif
}

}  // namespace slr
