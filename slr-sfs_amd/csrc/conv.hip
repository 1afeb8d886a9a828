// conv.hip -- 3x3 / stride 1 / pad 1 convolution of the partial-convolution decoder on the matrix
// cores (gfx950), fp32 in / fp32 out, as an implicit GEMM on v_mfma_f32_32x32x16_f16 with SPLIT
// operands, with the elementwise stages the reference wraps around every convolution fused into
// its prologue and epilogue (models/layers/partialconv2d.py:41-81,
// models/layers/blocks.py:173-248, models/layers/normalization.py:219-231; 95 % of a frame's time
// once the splat is fused).
//
// gfx950 has no TF32-like mode: fp32-input MFMA runs at the fp32 vector rate (157 TFLOP/s), 1/16
// of the f16 rate.  Every fp32 operand is therefore split into two halves, x = hi + lo with
// hi = f16(x), lo = f16(x - hi) (22 significant bits), and the product is formed from three f16
// MFMAs with fp32 accumulation:  x*w ~= hi*hi + hi*lo + lo*hi  (the dropped lo*lo term is
// 2^-22 relative).  Both operands are pre-scaled by powers of two so the lo halves stay normal
// f16 numbers; the inverse scale is applied (exactly) to the accumulators at the end.
// Measured error vs an fp64 convolution: 2-5e-6 on outputs of magnitude 4, the class of MIOpen's
// fp32 Winograd kernels (1-6e-6), at 3x their speed (tools/dev/convbench.py).
//
// Data flow of one workgroup (256 work-items = 4 waves, output block = 8 rows x 32 columns x
// (128 | 64 | 32) output channels of one sample):
//   * per chunk of 16 input channels the (8+2) x (32+2) input halo block is loaded from the NCHW
//     tensor (coalesced along x), run through the optional prologue relu(x*scale - shift)*mask,
//     split, and stored in LDS as [half][8-channel group][pixel] 16-byte vectors -- exactly the B
//     fragment of the MFMA (lane l: pixel l&31, channels 8*(l>>5)..+7), so fragment reads are
//     conflict-free ds_read_b128 and the 9 taps are just shifted pixel offsets; double-buffered:
//     the loads of chunk c+1 are in flight under the MFMAs of chunk c, one barrier per chunk;
//   * the pre-split weights are stored in fragment order
//     [co/32][ci/16][tap][half][ci group 2][co 32][ci 8]; a wave reads its A fragments straight from
//     global memory one tap ahead (1 KiB contiguous per fragment, L2-resident);
//   * a wave owns CPW 32-channel tiles x PT rows: CPW x PT accumulator tiles of 32x32 (128-channel
//     variant: wave wc = one channel tile x all 8 rows, 128 registers), 3 MFMAs per tile, tap and
//     chunk; the B fragments of the rows are read in two batches of four against the same A;
//   * epilogue on the accumulators: plain (+bias) or the partial-convolution one
//     (o = (raw*ratio + b)*um, then + residual or the next layer's relu(bn(.))*um), same operations
//     in the same order as csrc/pconv.hip.  The D layout (row = channel, column = pixel) makes every
//     accumulator register a pair of 128-byte row segments of the NCHW output.
#include "slr_common.hpp"

#include <atomic>
#include <mutex>
#include <type_traits>

namespace slr {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CV_W = 32, CV_H = 8;                 // output block
constexpr int CV_HW = CV_W + 2, CV_HH = CV_H + 2;  // input halo block
constexpr int CV_NPX = CV_HW * CV_HH;              // 340 halo pixels
constexpr int CV_THREADS = 256;
constexpr float CV_XSCALE = 64.0f;                 // default pre-scale of the activations before the split (2^6); per call:
                                                   // ConvArgs.xscale, a power of two in (0, 64] -- the split is exact-domain for
                                                   // |activation| < 65472 / xscale (1023 at 2^6, 65472 at 1), see stage_value
constexpr int CV_MAXCIN = 1024;                    // prologue scale/shift table in LDS

enum { PRE_NONE = 0, PRE_BN = 1, PRE_BN_MASK = 2, PRE_BN_NONZERO = 3 };

struct ConvArgs {
    const float *in;       // [N,Cin,H,W]
    const h8 *w;           // split weights in fragment order (see above), scaled by wscale
    float *out;            // [N,Cout,H,W]
    int N, Cin, Cout, H, W, tiles_x, nchunk;
    float unscale;         // 1 / (xscale * wscale)
    float xscale;          // pre-scale of the activations (power of two)
    // prologue: relu(x*pre_scale[c] - pre_shift[c]) * mask
    int pre;
    const float *pre_scale, *pre_shift;                // [Cin]
    const float *mask;     // [N,1,H,W] channel-uniform mask (prologue mask and/or mask plane of the update), or nullptr
    // epilogue
    const float *bias;     // [Cout] or nullptr
    int partial;           // partial-convolution epilogue (bias required)
    float mask_scale, winsize;   // conv(mask, ones) = box3x3(mask plane) * mask_scale; Cin * 9
    const float *residual, *next_scale, *next_shift;
    float *um_out;         // [N,1,H,W] or nullptr
    int out_b8;            // output in the channel-blocked layout [N, Cout/8, H, W, 8] (Cout % 8 == 0)
    int res_b8;            // ... and the residual as well (only together with out_b8)
    unsigned *sat;         // device counter: incremented by every wave that had to saturate an activation (see stage_value)
#ifdef SLR_TRACE
    long long *trace;      // development builds: 16 time stamps per workgroup of the Winograd kernel (tools/dev/trace_wino.py)
#endif
    int wino_groups;       // Winograd kernel (conv_wino.hpp): groups of 64 output channels (its grid.x carries blocks x groups)
    // the block's 1x1 skip branch as extra K chunks of this convolution (SKIP instantiations; blocks.py:243-248, :83-87):
    // out = epilogue(conv3x3(in)) + conv1x1(skip_in) (+ skip_bias)
    const float *skip_in;  // [N,skip_cin,H,W] channel-blocked, no prologue
    const h8 *skip_w;      // split 1x1 weights in fragment order (taps = 1), scaled by the skip's own wscale
    const float *skip_bias;
    int skip_cin, skip_nchunk;
    float skip_unscale;    // 1 / (xscale * skip wscale)
    // POOL instantiations: `out` is the 3x3 / stride 2 / pad 1 average pool of the result ("Down" blocks, blocks.py:196-199), channel-blocked;
    // the tile's last row / last column go to these side buffers for pool_fix_kernel: [N][tiles_y][Cout/8][W][8], [N][tiles_x][Cout/8][H][8]
    // UPS instantiations: `out` is the x2 bilinear up-sampling of the result ("Up" blocks, blocks.py:200-203); side buffers: first / last row
    // of every tile row, first / last column of every tile column: [N][tiles_y][2][Cout/8][W][8], [N][tiles_x][2][Cout/8][H][8]
    float *pool_row, *pool_col;
    int tiles_y, resample;   // resample: 0 none, 1 POOL, 2 UPS
    float *skip_out;       // conv3x3_few_kernel<.., SKP>: [N,Cout,H,W] = conv1x1(in) + skip_bias with skip_w = plain fp32 [Cin][4] weights
};

// padded channel counts of the weight buffer (shared by the split and the forward entry points)
__host__ __device__ inline int conv_cout_tile(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }
__host__ __device__ inline int conv_cout_pad(int Cout) { const int t = conv_cout_tile(Cout); return (Cout + t - 1) / t * t; }
__host__ __device__ inline int conv_cin_pad(int Cin) { return (Cin + 15) / 16 * 16; }

}  // namespace slr
#include "conv_few.hpp"
namespace slr {

// CPW: 32-channel output tiles per wave; WCO: waves along the output channels (workgroup covers
// 32*CPW*WCO channels); the other 4/WCO wave rows split the 8 block rows.
// INB8: the input tensor is channel-blocked, [N, Cin/8, H, W, 8] (what a previous call wrote with out_b8): the 8 channels of
// a staging item are 32 contiguous bytes = two 16-byte loads instead of eight 4-byte loads.  A VMEM instruction costs
// an in-order wave ~60-100 issue cycles; the 24 staging loads per chunk of the NCHW scheme are 14 % of the kernel.
// F32: the fp32 rung -- operands stay fp32, products on v_mfma_f32_32x32x2_f32 (fp32 multiply, fp32 accumulate: the arithmetic of the
// reference's convolutions, models/layers/partialconv2d.py:61-74; no pre-scale, no clamp, no magnitude limit) at the fp32 matrix
// rate (157 TFLOP/s, 1/16 of the f16 rate: 64 cycles per MFMA and SIMD).  Same workgroup shape, prologue, epilogue and layouts;
// the staged halo block is kept as [16 channels][halo pixel] floats (row stride 352: lanes 0-31 / 32-63 of a fragment read
// -- pixel l & 31 of channels 2k / 2k + 1 -- fall on disjoint banks), a wave's A fragments are 8 floats per lane and tap
// ([co tile][chunk][tap][lane][k pair]: two 16-byte loads), and one tap is 8 x PT MFMAs with ONE ds_read_b32 each: the loop is
// bound by the matrix pipe alone.
constexpr int CV_FSTR = 352;                       // F32: floats per channel row of the staged block
constexpr int CV_KSTR = 260;                       // F32, skip fills: floats per channel row (256 pixels; channels k and k + 8 -- lanes 0-31 / 32-63 of a fragment -- on disjoint banks)
// SKIP: the residual block's 1x1 skip convolution rides in this kernel (see the skip phase behind the main loop): the separate 1x1 kernel,
// the write of its result and the read of it as the residual are gone.
// POOL (with SKIP, 128-channel workgroup rows): the "Down" block's average pool in the epilogue -- see the end of the kernel.
// UPS (likewise): the "Up" block's x2 bilinear up-sampling in the epilogue.
template <int CPW, int WCO, bool PRE, bool INB8, bool F32 = false, bool SKIP = false, bool POOL = false, bool UPS = false>
__global__ __launch_bounds__(CV_THREADS, 2) void conv3x3_split_kernel(ConvArgs a) {
    constexpr int WPX = 4 / WCO, PT = CV_H / WPX;
    static_assert(!(POOL || UPS) || (SKIP && WCO == 4 && !(POOL && UPS)), "the resampling epilogues: a wave holds all 8 rows of its channel tile");
    static_assert(!F32 || CPW == 1, "the fp32 rung runs one 32-channel tile per wave");
    static_assert(!SKIP || (CPW == 1 && INB8), "the skip phase: one tile per wave, channel-blocked main input");
    constexpr int XRAW = F32 ? 2 * 16 * CV_FSTR * 4 : 2 * 2 * 2 * CV_NPX * 16, XSKIP = !SKIP ? 0 : SLR_CONV_SKIP_FILL * (F32 ? 16 * CV_KSTR * 4 : 2 * 2 * 256 * 16);
    __shared__ __attribute__((aligned(16))) unsigned char xraw[XRAW > XSKIP ? XRAW : XSKIP];
    h8 (*xs)[2][2][CV_NPX] = reinterpret_cast<h8 (*)[2][2][CV_NPX]>(xraw);             // [buffer][hi|lo][8-channel group][halo pixel]
    float (*xf)[16][CV_FSTR] = reinterpret_cast<float (*)[16][CV_FSTR]>(xraw);          // F32: [buffer][channel][halo pixel]
    __shared__ float mpl[CV_NPX];                  // mask plane over the halo block (0 outside the image)
    __shared__ float mplB[2][CV_NPX - 256];        // derived mask: per-group counts of the round-2 pixels
    __shared__ float4 pss4[2][CV_MAXCIN / 4];      // prologue scale / shift (x CV_XSCALE) per (padded) input channel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: keeps tile bases in SGPRs
    const int wc = wave % WCO, wp = wave / WCO;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int x0 = tx * CV_W, y0 = ty * CV_H;
    const int n = blockIdx.z;
    const int cot0 = (blockIdx.y * WCO + wc) * CPW;  // first 32-channel tile of this wave
    const int HW = a.H * a.W;
    const int nchunk = a.nchunk;
    const int cmax = a.Cin - 1;
    const float *inb = a.in + (size_t)n * a.Cin * HW;
    const int pre = a.pre;

    if (PRE) {
        float *pssw = reinterpret_cast<float *>(&pss4[0][0]);
        for (int i = tid; i < nchunk * 16; i += CV_THREADS) {       // padded channels: scale = shift = 0 -> 0
            pssw[i] = i < a.Cin ? a.pre_scale[i] * (F32 ? 1.0f : a.xscale) : 0.0f;
            pssw[CV_MAXCIN + i] = i < a.Cin ? a.pre_shift[i] * (F32 ? 1.0f : a.xscale) : 0.0f;
        }
    }

    // Staging of a chunk = 680 (halo pixel, 8-channel group) items in three rounds:
    //   round 0: pixel tid, channels 0-7;   round 1: pixel tid, channels 8-15;
    //   round 2: work-items < 168: pixel 256 + tid % 84, channels 8 * (tid / 84) .. +7.
    // Rounds 0/1 address a wave-uniform plane base + one 32-bit lane offset (no address VALU).
    const int gB = tid >= 84 ? 1 : 0;
    const int pB = 256 + tid - 84 * gB;
    const bool liveB = tid < 168;
    bool okA, okB;
    int offA, offB;
    {
        const int pr = tid / CV_HW, pc = tid - pr * CV_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        okA = (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        offA = okA ? gy * a.W + gx : 0;
    }
    {
        const int pr = pB / CV_HW, pc = pB - pr * CV_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        okB = liveB & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        offB = okB ? gy * a.W + gx : 0;
    }
    // per-item multiplier: the [N,1,H,W] mask value (1 without a mask), 0 for the zero padding outside
    // the image; without a prologue it also carries the 2^6 pre-scale of the split
    const float unit = (PRE || F32) ? 1.0f : a.xscale;
    const float mvA = (a.mask && okA) ? a.mask[(size_t)n * HW + offA] : 0.0f;
    const float mvB = (a.mask && okB) ? a.mask[(size_t)n * HW + offB] : 0.0f;
    if (a.mask) {                                      // mask plane of the halo block, for the 3x3 box sum of the epilogue
        mpl[tid] = mvA;
        if (tid < CV_NPX - 256) mpl[256 + tid] = mvB;
    }
    const float mA = okA ? (pre == PRE_BN_MASK ? mvA : unit) : 0.0f;
    const float mB = okB ? (pre == PRE_BN_MASK ? mvB : unit) : 0.0f;
    float cnt[3] = {0.0f, 0.0f, 0.0f};                 // derived mask: non-zero inputs per staging item, over all chunks
    const bool nonzero_mask = pre == PRE_BN_NONZERO;
    // R = staging round; c = chunk.  Channels past Cin re-read the last plane: finite values that meet
    // zero weights (and zero scale / shift with a prologue).
    const int c8max = (a.Cin >> 3) - 1;               // INB8: last 8-channel group
    auto load_round = [&](auto R, int c, float (&st)[8]) {
        if (INB8) {                                    // groups past Cin re-read the last one (zero weights)
            const int grp = min(c * 2 + (R.value < 2 ? R.value : gB), c8max);
            const float4 *q = reinterpret_cast<const float4 *>(inb) + ((size_t)grp * HW + (unsigned)(R.value < 2 ? offA : offB)) * 2;
            const float4 u = q[0], v = q[1];
            st[0] = u.x; st[1] = u.y; st[2] = u.z; st[3] = u.w;
            st[4] = v.x; st[5] = v.y; st[6] = v.z; st[7] = v.w;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (R.value < 2) {
                const float *pl = inb + (size_t)min(c * 16 + R.value * 8 + j, cmax) * HW;       // wave-uniform
                st[j] = pl[(unsigned)offA];
            } else {
                st[j] = inb[(unsigned)(min(c * 16 + gB * 8 + j, cmax) * HW + offB)];
            }
        }
    };
    // prologue (normalization.py:231, ReLU, partialconv2d.py:69; 2^6 folded into scale / shift, which
    // commutes with the roundings) + split + LDS store: straight-line code, no selects on validity.
    // In three pieces (begin / one value / finish) so the main loop can slot the values between MFMAs.
    struct Stage { int cb; float mk0, fresh, cnt; h8 hi, lo; float f[F32 ? 8 : 1]; };
    unsigned long long sat = 0ull;
    const float *pss = reinterpret_cast<const float *>(&pss4[0][0]);
    auto stage_begin = [&](auto R, int c, Stage &g) {
        g.cb = c * 16 + (R.value < 2 ? R.value * 8 : gB * 8);
        g.mk0 = R.value < 2 ? mA : mB;
        g.cnt = 0.0f;
    };
    auto stage_value = [&](int j, Stage &g, const float (&st)[8]) {
        const float x = st[j];
        float v;
        if (PRE) {           // scale / shift are read per value (2 broadcast LDS reads) rather than held in 16 registers
            const float mk = (nonzero_mask & (x == 0.0f)) ? 0.0f : g.mk0;
            v = fmaxf(x * pss[g.cb + j] - pss[CV_MAXCIN + g.cb + j], 0.0f) * mk;
            g.cnt += (g.cb + j <= cmax) ? mk : 0.0f;     // (x != 0) inside the image, real channels only: the derived
                                                         // mask's channel sum
        } else {
            v = x * g.mk0;
        }
        if (F32) { g.f[j] = v; return; }                     // the fp32 rung stages the value as it is
        // f16 range guard (one v_med3): the split is exact-domain for |activation| < 2^16 / 2^6 = 1023;
        // larger values saturate there instead of becoming inf -> NaN (post-BN activations are O(1..10^2)).
        // A saturated value makes the frame WRONG, not just inexact: it is counted (one compare into a lane mask
        // held in SGPRs, one atomic per affected wave at the end) and surfaced by slr_conv_saturation_count.
        v = __builtin_amdgcn_fmed3f(v, -65472.0f, 65472.0f);
        sat |= __ballot(fabsf(v) >= 65472.0f);            // tested AFTER the clamp (no second live copy of v); lane mask in SGPRs
        const _Float16 h = (_Float16)v;
        g.hi[j] = h;
        g.lo[j] = (_Float16)(v - (float)h);
    };
    auto stage_finish = [&](auto R, int buf, const Stage &g) {
        cnt[R.value] += g.cnt * g.fresh;
        const int dst = R.value == 0 ? tid : (R.value == 1 ? CV_NPX + tid : gB * CV_NPX + pB);
        if (F32) {
            const int ch0 = R.value < 2 ? R.value * 8 : gB * 8, px = R.value < 2 ? tid : pB;
            if (R.value < 2 || liveB) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[buf][ch0 + j][px] = g.f[j];
            }
            return;
        }
        if (R.value < 2 || liveB) {
            (&xs[buf][0][0][0])[dst] = g.hi;
            (&xs[buf][1][0][0])[dst] = g.lo;
        }
    };
    auto store_round = [&](auto R, int buf, int c, const float (&st)[8]) {
        Stage g;
        stage_begin(R, c, g);
        g.fresh = 1.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) stage_value(j, g, st);
        stage_finish(R, buf, g);
    };
    using R0 = std::integral_constant<int, 0>;
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;

    f16v acc[CPW][PT];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.0f;

    __syncthreads();                                   // pss4
    {
        float s0[8], s1[8], s2[8];                     // first chunk: all three rounds in flight at once
        load_round(R0{}, 0, s0);
        load_round(R1{}, 0, s1);
        load_round(R2{}, 0, s2);
        store_round(R0{}, 0, 0, s0);
        store_round(R1{}, 0, 0, s1);
        store_round(R2{}, 0, 0, s2);
    }
    __syncthreads();
    // DEEP (kernels with registers to spare: fewer than 8 rows per wave): a staging round's loads are issued five taps before the round is consumed
    // instead of two -- round 0 already in the previous chunk iteration -- in a register set per round.  A tap of these kernels is 6 / 12 MFMAs per wave:
    // two taps are less than the memory latency under load (the same kernels with every staging load redirected to a cached 4 KB ran 13 % faster).
    constexpr bool DEEP = SLR_CONV_DEEP_STAGE && WCO < 4 && !F32;
    float stx[DEEP ? 3 : 1][8];
    float (&st)[8] = stx[0];
    if (DEEP) load_round(R0{}, min(1, nchunk - 1), stx[0]);

    const int bcol = lane & 31, bgrp = lane >> 5;
    // A fragments (weights) come straight from global memory / L2, one tap ahead of their use.
    // Fragment (cot, chunk, tap, half) = 64 consecutive 16-byte vectors, lane l takes vector l.
    const size_t wtile = (size_t)nchunk * 9 * 2 * 64;            // vectors per 32-channel tile
    const h8 *wbase = a.w + (size_t)cot0 * wtile;                // wave-uniform
    h8 a_cur[CPW][2], a_nxt[CPW][2];                              // (F32: the same 8 registers hold the lane's 8 fp32 weights of a tap)
    auto load_a = [&](h8 (&dst)[CPW][2], int g /* chunk * 9 + tap */) {
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) {
            const h8 *q = wbase + ct * wtile + (size_t)g * 128;     // uniform base, lane offset
            if (F32) { dst[ct][0] = q[2u * (unsigned)lane]; dst[ct][1] = q[2u * (unsigned)lane + 1u]; continue; }   // 32 contiguous bytes per lane
            dst[ct][0] = q[(unsigned)lane];
            dst[ct][1] = q[(unsigned)lane + 64u];
        }
    };
    load_a(a_cur, 0);
    const int glast = nchunk * 9 - 1;
    // DEEP: the weight fragments run TWO taps ahead.  Memory returns in order: a fragment load issued behind a staging round's loads comes back with
    // them (memory latency, not L2 latency), and one tap of these kernels is shorter than that
    h8 a_nx2[DEEP ? CPW : 1][2];
    if (DEEP) { load_a(a_nxt, min(1, glast)); }
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        const int cn = min(c + 1, nchunk - 1);         // the chunk staged under this one's MFMAs (the last
                                                       // iteration re-stages its own chunk: harmless, and it keeps
                                                       // every load unconditional -> counted vmcnt waits)
        const h8 *xh = &xs[buf][0][bgrp][0], *xl = &xs[buf][1][bgrp][0];          // (split rung)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            // keeps the B fragments of different taps from being kept live together (the 36 distinct
            // ones of a chunk would take 144 registers and spill); LDS has the bandwidth to re-read them
            asm volatile("" ::: "memory");
            // staging of the next chunk, one round per three taps: 8 registers in flight instead of 24.
            // A round's loads are issued at the top of taps 0 / 3 / 6 (behind the weight loads) and consumed
            // (prologue, split, LDS store) in taps 2 / 5 / 8, where that VALU work is interleaved with
            // the tap's MFMAs (below).
            if (DEEP) load_a(a_nx2, min(c * 9 + tap + 2, glast));
            else load_a(a_nxt, min(c * 9 + tap + 1, glast));
            // AFTER the weight loads: memory returns in order, so a staging load (HBM latency) issued in
            // front of an A load (L2 latency) would make the next tap wait for HBM
            if (DEEP) {
                if (tap == 0) load_round(R1{}, cn, stx[DEEP ? 1 : 0]);
                if (tap == 3) load_round(R2{}, cn, stx[DEEP ? 2 : 0]);
                if (tap == 6) load_round(R0{}, min(c + 2, nchunk - 1), stx[0]);
            } else {
                if (tap == 0) load_round(R0{}, cn, st);
                if (tap == 3) load_round(R1{}, cn, st);
                if (tap == 6) load_round(R2{}, cn, st);
            }
            __builtin_amdgcn_sched_barrier(0);         // loads are issued HERE, a whole tap ahead of their use
            const bool stage_tap = tap == 2 || tap == 5 || tap == 8;
            Stage sg;
            sg.fresh = c + 1 < nchunk ? 1.0f : 0.0f;   // the last iteration re-stages its own chunk: not counted twice
            if (stage_tap) {
                if (tap == 2) stage_begin(R0{}, cn, sg);
                if (tap == 5) stage_begin(R1{}, cn, sg);
                if (tap == 8) stage_begin(R2{}, cn, sg);
            }
            // The three partial products (small terms first), each over all CPW x PT accumulator tiles:
            // consecutive MFMAs never touch the same accumulator.  In a staging tap the 8 values of the
            // round are converted BETWEEN groups of MFMAs (order pinned by scheduling barriers), so that
            // VALU work issues in the shadow of the matrix pipe instead of in front of it.
            // B fragments are read in batches of <= 4 pixel tiles (32 registers): a wave with 8 rows
            // (WCO = 4) makes two passes with the same A fragments.
            constexpr int PB = PT > 4 ? 4 : PT, NB = PT / PB;
            constexpr int NM = 3 * CPW * PT, NMB = 3 * CPW * PB;
            if constexpr (F32) {
                // 8 k-pairs x PT rows: MFMA (kp, row) multiplies the tap's weights of input channels 2kp, 2kp + 1 (lanes 0-31 / 32-63) with
                // the row's 32 pixels of those two channels; consecutive MFMAs never touch the same accumulator
                typedef float f8v __attribute__((ext_vector_type(8)));
                union AW { h8 h[2]; f8v f; } aw;
                aw.h[0] = a_cur[0][0]; aw.h[1] = a_cur[0][1];
                constexpr int NMF = 8 * PT;
#pragma unroll
                for (int kp = 0; kp < 8; ++kp) {
                    float bv[PT];
#pragma unroll
                    for (int k = 0; k < PT; ++k) bv[k] = xf[buf][2 * kp + bgrp][(wp * PT + k + kh) * CV_HW + kw + bcol];
#pragma unroll
                    for (int k = 0; k < PT; ++k) {
                        acc[0][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.f[kp], bv[k], acc[0][k], 0, 0, 0);
                        if (stage_tap) {
                            const int i = kp * PT + k;
                            const int j0 = i * 8 / NMF, j1 = (i + 1) * 8 / NMF;
                            if (j1 > j0) {
#pragma unroll
                                for (int j = j0; j < j1; ++j) stage_value(j, sg, stx[DEEP ? tap / 3 : 0]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int hb = 0; hb < NB; ++hb) {
                h8 bh[PB], bl[PB];
#pragma unroll
                for (int k = 0; k < PB; ++k) {
                    const int p = (wp * PT + hb * PB + k + kh) * CV_HW + kw + bcol;
                    bh[k] = xh[p];
                    bl[k] = xl[p];
                }
#pragma unroll
                for (int ib = 0; ib < NMB; ++ib) {
                    const int part = ib / (CPW * PB), k = (ib / CPW) % PB, ct = ib % CPW;
                    const int pt = hb * PB + k, i = hb * NMB + ib;
                    const h8 av = part == 0 ? a_cur[ct][1] : a_cur[ct][0];
                    const h8 bv = part == 1 ? bl[k] : bh[k];
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[ct][pt], 0, 0, 0);
                    if (stage_tap) {
                        const int j0 = i * 8 / NM, j1 = (i + 1) * 8 / NM;
                        if (j1 > j0) {
#pragma unroll
                            for (int j = j0; j < j1; ++j) stage_value(j, sg, stx[DEEP ? tap / 3 : 0]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            if (stage_tap) {
                if (tap == 2) stage_finish(R0{}, buf ^ 1, sg);
                if (tap == 5) stage_finish(R1{}, buf ^ 1, sg);
                if (tap == 8) stage_finish(R2{}, buf ^ 1, sg);
            }
#pragma unroll
            for (int ct = 0; ct < CPW; ++ct) {
                a_cur[ct][0] = a_nxt[ct][0]; a_cur[ct][1] = a_nxt[ct][1];
                if (DEEP) { a_nxt[ct][0] = a_nx2[ct][0]; a_nxt[ct][1] = a_nx2[ct][1]; }
            }
        }
        __syncthreads();
    }

    if (pre == PRE_BN_NONZERO) {                       // mask plane = channel sum of (x != 0)  (architectures.py:369,
        mpl[tid] = cnt[0] + cnt[1];                    // partialconv2d.py:61 with a per-element mask)
        if (liveB) mplB[gB][pB - 256] = cnt[2];
        __syncthreads();
        if (tid < CV_NPX - 256) mpl[256 + tid] = mplB[0][tid] + mplB[1][tid];
    }
    __syncthreads();                                   // mpl complete (written before the main loop otherwise)

    if constexpr (SKIP) {
        // ---- skip phase.  The accumulators take the 3x3 convolution's epilogue NOW (the operations of the epilogue below, in its order) and
        // go on, in the scale of the skip operands, as the accumulators of the block's 1x1 skip convolution:
        //   out = ((raw*ratio + b)*um  |  raw + b)  +  conv1x1(skip_in)  (+ skip_bias)          blocks.py:243-248 / :83-87
        // The change of scale is a multiplication by a power of two (exact).  A skip chunk (16 input channels) meets ONE tap -- 3 MFMAs per
        // row -- so this phase is bound by the latency of its loads, not by the matrix pipe: the chunks are staged SK at a time (only the
        // block's own 256 pixels, no halo: 16 KiB per chunk in the main loop's buffers), the loads of a fill are in flight under the
        // epilogue arithmetic resp. the previous fill's MFMAs.  (First form, one chunk per barrier with the halo staging of the main loop:
        // the four fused kernels of a decoder frame cost 18 % more than without their skips; HISTORY.md.)
        constexpr int SK = SLR_CONV_SKIP_FILL;
        static_assert(SK * (F32 ? 16 * CV_KSTR * 4 : 2 * 2 * 256 * 16) <= (int)sizeof(xraw), "skip fills live in the staging buffers");
        h8 (*xk)[2][2][256] = reinterpret_cast<h8 (*)[2][2][256]>(xraw);      // [chunk of the fill][hi|lo][8-channel group][pixel of the block]
        float (*xkf)[16][CV_KSTR] = reinterpret_cast<float (*)[16][CV_KSTR]>(xraw);   // F32
        const int ns = a.skip_nchunk, sc8max = (a.skip_cin >> 3) - 1;
        const int spr = tid >> 5, spc = tid & 31;
        const bool sok = (y0 + spr < a.H) & (x0 + spc < a.W);
        const float smk = sok ? a.xscale : 0.0f;                             // validity x the pre-scale of the split
        const float4 *sq = reinterpret_cast<const float4 *>(a.skip_in + (size_t)n * a.skip_cin * HW) + (size_t)(sok ? (y0 + spr) * a.W + x0 + spc : 0) * 2;
        float4 sv[SK][4];
        auto sload = [&](int s0) {                     // chunks s0 .. s0 + SK - 1 (past the last: re-read it, unused -- every load unconditional)
#pragma unroll
            for (int k = 0; k < SK; ++k)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 *q = sq + (size_t)min(min(s0 + k, ns - 1) * 2 + g, sc8max) * HW * 2;      // (uniform plane base)
                    sv[k][2 * g] = q[0];
                    sv[k][2 * g + 1] = q[1];
                }
        };
        auto sstore = [&]() {
#pragma unroll
            for (int k = 0; k < SK; ++k)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float x8[8] = {sv[k][2 * g].x, sv[k][2 * g].y, sv[k][2 * g].z, sv[k][2 * g].w,
                                         sv[k][2 * g + 1].x, sv[k][2 * g + 1].y, sv[k][2 * g + 1].z, sv[k][2 * g + 1].w};
                    if constexpr (F32) {               // fp32 rung: the values as they are, [chunk of the fill][channel][pixel of the block]
#pragma unroll
                        for (int j = 0; j < 8; ++j) xkf[k][8 * g + j][tid] = x8[j] * smk;
                        continue;
                    }
                    h8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = __builtin_amdgcn_fmed3f(x8[j] * smk, -65472.0f, 65472.0f);         // (the range guard of stage_value)
                        sat |= __ballot(fabsf(v) >= 65472.0f);
                        const _Float16 h = (_Float16)v;
                        hi[j] = h;
                        lo[j] = (_Float16)(v - (float)h);
                    }
                    xk[k][0][g][tid] = hi;
                    xk[k][1][g][tid] = lo;
                }
        };
        sload(0);
        {
            const bool partial = a.partial != 0, has_bias = a.bias != nullptr;
            const float unscale = a.unscale, rescale = 1.0f / a.skip_unscale;
            const float mscale = a.mask_scale, winsize = a.winsize;
            float eb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) eb[r] = has_bias ? a.bias[min(cot0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.Cout - 1)] : 0.0f;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                if (partial) {
                    float box = 0.0f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) box += mpl[(wp * PT + pt + dy) * CV_HW + (lane & 31) + dx];
                    const float u = box * mscale;
                    const float um = fminf(fmaxf(u, 0.0f), 1.0f);
                    const float ratio = (1.0f / (u + 1e-8f)) * winsize * um;
                    const int oy = y0 + wp * PT + pt, ox = x0 + (lane & 31);
                    if (ox < a.W && oy < a.H && a.um_out && blockIdx.y == 0 && wc == 0 && lane < 32)
                        a.um_out[(size_t)n * HW + (size_t)oy * a.W + ox] = um;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][pt][r] = (((acc[0][pt][r] * unscale) * ratio + eb[r]) * um) * rescale;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][pt][r] = (acc[0][pt][r] * unscale + eb[r]) * rescale;
                }
            }
        }
        const h8 *swbase = a.skip_w + (size_t)cot0 * ((size_t)ns * 128);           // fragment (tile, chunk, half): 64 vectors
        for (int s0 = 0; s0 < ns; s0 += SK) {
            sstore();
            h8 sa[SK][2];
#pragma unroll
            for (int k = 0; k < SK; ++k) {
                const h8 *q = swbase + (size_t)min(s0 + k, ns - 1) * 128;
                if (F32) { sa[k][0] = q[2u * (unsigned)lane]; sa[k][1] = q[2u * (unsigned)lane + 1u]; continue; }     // the lane's 8 fp32 weights
                sa[k][0] = q[(unsigned)lane]; sa[k][1] = q[(unsigned)lane + 64u];
            }
            __syncthreads();
            sload(s0 + SK);                            // the next fill, in flight under this one's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < SK; ++k) {
                if (s0 + k >= ns) break;               // (uniform)
                if constexpr (F32) {
                    // MFMA kp: the weights of input channels kp (lanes 0-31) and 8 + kp (lanes 32-63) -- slr_conv1x1_f32_weights' pairing --
                    // with the row's 32 pixels of those channels
                    typedef float f8v __attribute__((ext_vector_type(8)));
                    union AW { h8 h[2]; f8v f; } aw;
                    aw.h[0] = sa[k][0]; aw.h[1] = sa[k][1];
#pragma unroll
                    for (int kp = 0; kp < 8; ++kp) {
                        float bv[PT];
#pragma unroll
                        for (int kk = 0; kk < PT; ++kk) bv[kk] = xkf[k][kp + 8 * (lane >> 5)][(wp * PT + kk) * 32 + (lane & 31)];
#pragma unroll
                        for (int kk = 0; kk < PT; ++kk) acc[0][kk] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.f[kp], bv[kk], acc[0][kk], 0, 0, 0);
                    }
                    continue;
                }
                const h8 *xh = &xk[k][0][lane >> 5][0], *xl = &xk[k][1][lane >> 5][0];
                constexpr int PB = PT > 4 ? 4 : PT, NB = PT / PB;
#pragma unroll
                for (int hb = 0; hb < NB; ++hb) {
                    h8 bh[PB], bl[PB];
#pragma unroll
                    for (int kk = 0; kk < PB; ++kk) {
                        const int p = (wp * PT + hb * PB + kk) * 32 + (lane & 31);
                        bh[kk] = xh[p];
                        bl[kk] = xl[p];
                    }
#pragma unroll
                    for (int part = 0; part < 3; ++part)
#pragma unroll
                        for (int kk = 0; kk < PB; ++kk)
                            acc[0][hb * PB + kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(part == 0 ? sa[k][1] : sa[k][0], part == 1 ? bl[kk] : bh[kk],
                                                                                          acc[0][hb * PB + kk], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }

    if (!F32 && a.sat && sat != 0ull && lane == 0) atomicAdd(a.sat, 1u);   // an activation left the f16 range (stage_value)

    if constexpr (POOL) {
        // ---- pooling epilogue ("Down" blocks: the full-resolution result has one reader, nn.AvgPool2d(3, 2, 1) -- it is never written).
        // A wave holds all 8 rows x 32 columns of its 32 channels: pooled pixel (i, j) of the tile = rows 2i-1 .. 2i+1 (register index) x
        // columns 2j-1 .. 2j+1 (neighbouring lanes) -- 4 x 16 pooled pixels, of which the first row and the first column lack the row above /
        // the column left of the tile.  Those get their partial sums here; the tile's last row and last column go to side buffers and
        // pool_fix_kernel adds what is missing.  Sums are formed rows first, then columns (slr_avgpool3x3s2: columns first): same 9 terms, the
        // result differs from the two-kernel form by fp32 rounding only.
        const float unscale = a.skip_unscale;
        const int ox = x0 + bcol;
        const bool xin = ox < a.W;
        const int OH = (a.H - 1) / 2 + 1, OW = (a.W - 1) / 2 + 1, C8 = a.Cout >> 3;
        float sb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sb[r] = a.skip_bias ? a.skip_bias[min(cot0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp, a.Cout - 1)] : 0.0f;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const bool ok = xin & (y0 + pt < a.H);                     // pixels outside the image are the pool's zero padding
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][pt][r] = ok ? acc[0][pt][r] * unscale + sb[r] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c8 = cot0 * 4 + q;
            if (c8 * 8 + 4 * bgrp >= a.Cout) continue;
            if (xin && y0 + 7 < a.H) {                                 // last row of the tile
                float *rb = a.pool_row + ((((size_t)n * a.tiles_y + ty) * C8 + c8) * a.W + ox) * 8 + 4 * bgrp;
                *reinterpret_cast<float4 *>(rb) = make_float4(acc[0][7][4 * q], acc[0][7][4 * q + 1], acc[0][7][4 * q + 2], acc[0][7][4 * q + 3]);
            }
            if (bcol == 31 && xin) {                                   // last column
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    if (y0 + pt < a.H) {
                        float *cb = a.pool_col + ((((size_t)n * a.tiles_x + tx) * C8 + c8) * a.H + y0 + pt) * 8 + 4 * bgrp;
                        *reinterpret_cast<float4 *>(cb) = make_float4(acc[0][pt][4 * q], acc[0][pt][4 * q + 1], acc[0][pt][4 * q + 2], acc[0][pt][4 * q + 3]);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float hsum[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = i > 0 ? (acc[0][2 * i - 1][r] + acc[0][2 * i][r]) + acc[0][2 * i + 1][r] : acc[0][0][r] + acc[0][1][r];
                const float left = __shfl_up(v, 1, 32), right = __shfl_down(v, 1, 32);
                hsum[r] = (((bcol > 0 ? left : 0.0f) + v) + (bcol < 31 ? right : 0.0f)) * (1.0f / 9.0f);
            }
            const int py = (y0 >> 1) + i, px = (x0 + bcol) >> 1;
            if ((bcol & 1) == 0 && xin && y0 + 2 * i < a.H) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c8 = cot0 * 4 + q;
                    if (c8 * 8 + 4 * bgrp < a.Cout)
                        *reinterpret_cast<float4 *>(&a.out[((((size_t)n * C8 + c8) * OH + py) * OW + px) * 8 + 4 * bgrp]) =
                            make_float4(hsum[4 * q], hsum[4 * q + 1], hsum[4 * q + 2], hsum[4 * q + 3]);
                }
            }
        }
        return;
    }

    if constexpr (UPS) {
        // ---- up-sampling epilogue ("Up" blocks: nn.Upsample(scale_factor=2, mode='bilinear') of the result; the low-resolution result is never
        // written).  The tile's 8 x 32 pixels give the 16 x 64 output pixels below them; all but the first / last row and column of those
        // need values of this tile only: rows are registers, columns neighbouring lanes.  Expression and order of slr_upsample_bilinear2x
        // (ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11), weights 0.25 / 0.75, clamped neighbours at the image border):
        // bit-identical to the two-kernel form.  The border rows / columns of the 16 x 64 block are written by upsample_fix_kernel from the
        // side buffers (first / last row and column of every tile).
        const float unscale = a.skip_unscale;
        const int ox = x0 + bcol;
        const bool xin = ox < a.W, rclamp = ox + 1 >= a.W;
        const int OH = 2 * a.H, OW = 2 * a.W, C8 = a.Cout >> 3;
        float sb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sb[r] = a.skip_bias ? a.skip_bias[min(cot0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp, a.Cout - 1)] : 0.0f;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][pt][r] = acc[0][pt][r] * unscale + sb[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c8 = cot0 * 4 + q;
            if (c8 * 8 + 4 * bgrp >= a.Cout) continue;
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const int pt = slot * 7;
                if (xin && y0 + pt < a.H) {
                    float *rb = a.pool_row + (((((size_t)n * a.tiles_y + ty) * 2 + slot) * C8 + c8) * a.W + ox) * 8 + 4 * bgrp;
                    *reinterpret_cast<float4 *>(rb) = make_float4(acc[0][pt][4 * q], acc[0][pt][4 * q + 1], acc[0][pt][4 * q + 2], acc[0][pt][4 * q + 3]);
                }
            }
            if ((bcol == 0 || bcol == 31) && xin) {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    if (y0 + pt < a.H) {
                        float *cb = a.pool_col + (((((size_t)n * a.tiles_x + tx) * 2 + (bcol ? 1 : 0)) * C8 + c8) * a.H + y0 + pt) * 8 + 4 * bgrp;
                        *reinterpret_cast<float4 *>(cb) = make_float4(acc[0][pt][4 * q], acc[0][pt][4 * q + 1], acc[0][pt][4 * q + 2], acc[0][pt][4 * q + 3]);
                    }
            }
        }
#pragma unroll
        for (int pt = 1; pt < PT; ++pt)
            if (y0 + pt >= a.H) {                                      // (uniform) rows below the image: the clamped neighbour = the last row
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][pt][r] = acc[0][pt - 1][r];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c8 = cot0 * 4 + q;
            float hE[PT][4], hO[PT][4];                                // the rows' horizontal interpolations at output columns 2x, 2x + 1
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[0][pt][4 * q + e];
                    const float left = __shfl_up(v, 1, 32), rgt = __shfl_down(v, 1, 32);
                    hE[pt][e] = 0.25f * left + 0.75f * v;
                    hO[pt][e] = 0.75f * v + 0.25f * (rclamp ? v : rgt);
                }
            if (!xin || c8 * 8 + 4 * bgrp >= a.Cout) continue;
            float *ob = a.out + (((size_t)n * C8 + c8) * OH * OW + (size_t)(2 * y0) * OW + 2 * ox) * 8 + 4 * bgrp;
#pragma unroll
            for (int hr = 1; hr < 2 * PT - 1; ++hr) {
                const int ra = (hr & 1) ? hr >> 1 : (hr >> 1) - 1, rb = ra + 1;
                const float l0 = (hr & 1) ? 0.75f : 0.25f, l1 = (hr & 1) ? 0.25f : 0.75f;
                if (y0 + (hr >> 1) >= a.H) continue;                   // (uniform)
                float *orow = ob + (size_t)hr * OW * 8;
                if (bcol > 0)
                    *reinterpret_cast<float4 *>(orow) = make_float4(l0 * hE[ra][0] + l1 * hE[rb][0], l0 * hE[ra][1] + l1 * hE[rb][1],
                                                                    l0 * hE[ra][2] + l1 * hE[rb][2], l0 * hE[ra][3] + l1 * hE[rb][3]);
                if (bcol < 31)
                    *reinterpret_cast<float4 *>(orow + 8) = make_float4(l0 * hO[ra][0] + l1 * hO[rb][0], l0 * hO[ra][1] + l1 * hO[rb][1],
                                                                        l0 * hO[ra][2] + l1 * hO[rb][2], l0 * hO[ra][3] + l1 * hO[rb][3]);
            }
        }
        return;
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel).
    // Work-items outside the image / channels past Cout are clamped for the loads and skipped for
    // the stores; all loads of a tile are issued before its first store.
    // The optional stages are uniform branches around whole 16-register blocks (never per element),
    // and every block issues all its loads before its arithmetic.
    // (SKIP: the 3x3 epilogue has been applied in front of the skip phase; what is left is the change of scale and the skip's bias)
    const float *ebias = SKIP ? a.skip_bias : a.bias;
    // (has_res stays a run-time test in the SKIP kernels although conv_set_skip admits no residual: with a constant there the
    // compiler keeps the transposed tile of the vector store path in scratch memory)
    const bool partial = !SKIP && a.partial != 0, has_bias = ebias != nullptr, has_res = a.residual != nullptr,
               has_next = !SKIP && a.next_scale != nullptr;
    float *outp = a.out;
    // Vector path of the stores: the finished 32-channel x 32-pixel tile takes a round trip through the wave's own
    // LDS scratch (the staging buffers are free now) and comes back as [channel][4 consecutive pixels] per lane, so
    // the residual is read and the result written with 16-byte accesses: 4 store instructions per tile instead
    // of 16 (a VMEM instruction costs an in-order wave ~60-100 issue cycles).  Needs whole 32-pixel rows inside
    // the image and 16-byte aligned rows; otherwise the 4-byte path below.
    const bool vec = !a.out_b8 && (a.W % 4 == 0) && (x0 + CV_W <= a.W) &&
                     !(((uintptr_t)a.out | (uintptr_t)a.residual) & 15);
    constexpr int SCR_STRIDE = 36;                     // floats per channel row: 16-byte aligned, conflict-free
    float *scr = reinterpret_cast<float *>(xraw) + wave * (32 * SCR_STRIDE);
    const int ox = x0 + bcol;
    const bool xin_img = ox < a.W;
    const int cout1 = a.Cout - 1;
    const float mscale = a.mask_scale, winsize = a.winsize, unscale = SKIP ? a.skip_unscale : a.unscale;
#pragma unroll
    for (int ct = 0; ct < CPW; ++ct) {
        int co[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) co[r] = (cot0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp;
        float eb[16], esc[16], esh[16];
        if (has_bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) eb[r] = ebias[min(co[r], cout1)];
        }
        if (has_next) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { esc[r] = a.next_scale[min(co[r], cout1)]; esh[r] = a.next_shift[min(co[r], cout1)]; }
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int oy = y0 + wp * PT + pt;
            const bool ok = xin_img & (oy < a.H);
            const size_t pix = ok ? (size_t)oy * a.W + ox : 0;
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = acc[ct][pt][r] * unscale;
            if (partial) {
                float box = 0.0f;                                  // conv(mask, ones): 3x3 box sum, zero padded (:61)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) box += mpl[(wp * PT + pt + dy) * CV_HW + bcol + dx];
                const float u = box * mscale;                      // exact small integers in fp32
                const float um = fminf(fmaxf(u, 0.0f), 1.0f);
                const float ratio = (1.0f / (u + 1e-8f)) * winsize * um;       // torch: scalar / tensor = reciprocal * scalar
                if (ok && a.um_out && ct == 0 && blockIdx.y == 0 && wc == 0 && bgrp == 0)
                    a.um_out[(size_t)n * HW + pix] = um;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = (o[r] * ratio + eb[r]) * um;                   // :72-74
                if (has_res && !vec && !a.out_b8) {                                                 // blocks.py:248
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = a.residual[((size_t)n * a.Cout + min(co[r], cout1)) * HW + pix];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] += rv[r];
                }
                if (has_next) {                                                                     // blocks.py:233-236
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = fmaxf(o[r] * esc[r] - esh[r], 0.0f) * um;
                }
            } else {                                       // plain convolution: (+ bias) (+ residual, blocks.py:87)
                if (has_bias) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] += eb[r];
                }
                if (has_res && !vec && !a.out_b8) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = a.residual[((size_t)n * a.Cout + min(co[r], cout1)) * HW + pix];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] += rv[r];
                }
            }
            if (a.out_b8) {
                // channel-blocked output: register group q = r >> 2 holds channels 8q + 4*bgrp + (0..3) of this pixel,
                // i.e. 16 contiguous bytes of [N, Cout/8, H, W, 8]: 4 stores per tile, no transpose.  The residual
                // (last operation of both epilogues) is read in its own layout: blocked (16 bytes) or NCHW.
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c8 = (cot0 + ct) * 4 + q;
                    const bool live = ok && c8 * 8 + 4 * bgrp < a.Cout;
                    const size_t bidx = (((size_t)n * (a.Cout >> 3) + (live ? c8 : 0)) * HW + pix) * 8 + 4 * bgrp;
                    float4 v = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                    if (has_res) {
                        float4 rv;
                        if (a.res_b8) {
                            rv = *reinterpret_cast<const float4 *>(&a.residual[bidx]);
                        } else {
                            const size_t ridx = ((size_t)n * a.Cout + (live ? c8 * 8 + 4 * bgrp : 0)) * HW + pix;
                            rv = make_float4(a.residual[ridx], a.residual[ridx + HW], a.residual[ridx + 2 * (size_t)HW], a.residual[ridx + 3 * (size_t)HW]);
                        }
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    if (live) *reinterpret_cast<float4 *>(&outp[bidx]) = v;
                }
            } else if (vec) {
                // (the residual add is the LAST operation of both epilogues, so it moves behind the transpose unchanged)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * bgrp) * SCR_STRIDE + bcol] = o[r];
                __builtin_amdgcn_wave_barrier();
                const int quad = lane & 7;
                float4 v[4], rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4 *>(&scr[((lane >> 3) + 8 * k) * SCR_STRIDE + quad * 4]);
                __builtin_amdgcn_wave_barrier();
                const bool rowok = oy < a.H;
                size_t vidx[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cv = min((cot0 + ct) * 32 + (lane >> 3) + 8 * k, cout1);
                    vidx[k] = ((size_t)n * a.Cout + cv) * HW + (rowok ? (size_t)oy * a.W + x0 + quad * 4 : 0);
                }
                if (has_res) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) rv[k] = *reinterpret_cast<const float4 *>(&a.residual[vidx[k]]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[k].x += rv[k].x; v[k].y += rv[k].y; v[k].z += rv[k].z; v[k].w += rv[k].w; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (rowok && (cot0 + ct) * 32 + (lane >> 3) + 8 * k <= cout1) *reinterpret_cast<float4 *>(&outp[vidx[k]]) = v[k];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (ok && co[r] <= cout1) outp[((size_t)n * a.Cout + co[r]) * HW + pix] = o[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution (the skip branch of the residual blocks, models/layers/blocks.py:192-193,243-247)
// on the same split-f16 MFMA arithmetic.  HBM-bound (one read of the input, one write of the
// output) and free of LDS and barriers: the B fragment of v_mfma_f32_32x32x16_f16 (lane l: pixel
// l&31, channels 8*(l>>5)..+7) is exactly what 8 coalesced plane loads per lane deliver, so a wave
// converts its 32 pixels x 16 channels in registers and multiplies them with all NCT 32-channel
// weight tiles (A fragments from global memory / L2).  Input loads run two chunks ahead.
constexpr int C1_TILES = 1;                        // 32-pixel tiles per wave (streaming several was measured slower:
                                                   // the stores of a tile share vmcnt with the next tile's loads)
template <int NCT, bool INB8, bool F32 = false>
__global__ __launch_bounds__(256) void conv1x1_split_kernel(const float *__restrict__ in, const h8 *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            int Cin, int Cout, int HW, int nchunk, float unscale, float xscale,
                                                            int out_b8, unsigned *__restrict__ sat_count) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.z;
    const int cot0 = blockIdx.y * NCT;
    const int pcol = lane & 31, grp = lane >> 5;
    // (tile, chunk) is ONE software-pipelined sequence g = tile * nchunk + chunk
    const int p0 = ((blockIdx.x * 4 + wave) * C1_TILES) * 32 + pcol;
    const int cmax = Cin - 1;
    const int gend = C1_TILES * nchunk;
    const float *inb = in + (size_t)n * Cin * HW;
    auto load_x = [&](float (&x)[8], int g) {          // channels past Cin re-read the last plane (zero weights)
        g = min(g, gend - 1);
        const int t = g / nchunk, c = g - t * nchunk;
        const int p = p0 + t * 32;
        const unsigned poff = p < HW ? p : 0;
        if (INB8) {                                    // channel-blocked input: the lane's 8 channels are 32 contiguous bytes
            const float4 *q4 = reinterpret_cast<const float4 *>(inb) + ((size_t)min(c * 2 + grp, (Cin >> 3) - 1) * HW + poff) * 2;
            const float4 u = q4[0], v = q4[1];
            x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = inb[(size_t)min(c * 16 + grp * 8 + j, cmax) * HW + poff];
    };
    const h8 *wb = w + (size_t)cot0 * nchunk * 128 + lane;      // fragment (tile, chunk, half): 64 vectors
    auto load_a = [&](h8 (&d)[NCT][2], int g) {
        const int c = g % nchunk;
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const h8 *q = wb + ((size_t)t * nchunk + c) * 128;
            if (F32) { d[t][0] = q[lane]; d[t][1] = q[lane + 1]; continue; }     // (wb + lane + lane: the lane's 8 fp32 weights, 32 contiguous bytes)
            d[t][0] = q[0];
            d[t][1] = q[64];
        }
    };
    // epilogue constants
    const int cout1 = Cout - 1;
    const bool has_bias = bias != nullptr;
    const float *bp = has_bias ? bias : in;
    f16v acc[NCT];
    float x0[8], x1[8], x2[8];
    h8 a_cur[NCT][2], a_nxt[NCT][2];
    bool sat = false;                                  // an activation left the f16 range of the split (see conv3x3_split_kernel)
    load_x(x0, 0);
    load_a(a_cur, 0);
    load_x(x1, 1);
    for (int tile = 0, g = 0; tile < C1_TILES; ++tile) {
        const int p = p0 + tile * 32;
        const bool ok = p < HW;
        const float okf = ok ? xscale : 0.0f;
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        for (int c = 0; c < nchunk; ++c, ++g) {
            // everything issued here is for LATER chunks and unconditional, so the waits below are counted:
            // the weights of chunk g+1 and the input of chunk g+2 stay in flight under this chunk's MFMAs
            load_a(a_nxt, g + 1);
            load_x(x2, g + 2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (F32) {
                // fp32 rung: MFMA kp multiplies the weights of input channels 8 * (lane >> 5) + kp with this lane's pixel of that channel
                typedef float f8v __attribute__((ext_vector_type(8)));
                const float ok1 = ok ? 1.0f : 0.0f;
#pragma unroll
                for (int kp = 0; kp < 8; ++kp)
#pragma unroll
                    for (int t = 0; t < NCT; ++t) {
                        union AW { h8 h[2]; f8v f; } aw;
                        aw.h[0] = a_cur[t][0]; aw.h[1] = a_cur[t][1];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw.f[kp], x0[kp] * ok1, acc[t], 0, 0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j) { x0[j] = x1[j]; x1[j] = x2[j]; }
#pragma unroll
                for (int t = 0; t < NCT; ++t) { a_cur[t][0] = a_nxt[t][0]; a_cur[t][1] = a_nxt[t][1]; }
                continue;
            }
            h8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = __builtin_amdgcn_fmed3f(x0[j] * okf, -65472.0f, 65472.0f);
                sat |= fabsf(v) >= 65472.0f;                   // (the same test as the 3x3 kernel: after the clamp)
                const _Float16 h = (_Float16)v;
                bh[j] = h;
                bl[j] = (_Float16)(v - (float)h);
            }
#pragma unroll
            for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[t][1], bh, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[t][0], bl, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NCT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[t][0], bh, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { x0[j] = x1[j]; x1[j] = x2[j]; }
#pragma unroll
            for (int t = 0; t < NCT; ++t) { a_cur[t][0] = a_nxt[t][0]; a_cur[t][1] = a_nxt[t][1]; }
        }
        // straight-line epilogue: clamped bias loads first, then masked stores
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            float b[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = bp[min((cot0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * grp, cout1)];
                b[r] = has_bias ? v : 0.0f;
            }
            if (out_b8) {                              // 4 channels of one 8-group per register group: 16-byte stores
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c8 = (cot0 + t) * 4 + q;
                    if (ok && c8 * 8 + 4 * grp < Cout)
                        *reinterpret_cast<float4 *>(&out[(((size_t)n * (Cout >> 3) + c8) * HW + p) * 8 + 4 * grp]) =
                            make_float4(acc[t][4 * q] * unscale + b[4 * q], acc[t][4 * q + 1] * unscale + b[4 * q + 1],
                                        acc[t][4 * q + 2] * unscale + b[4 * q + 2], acc[t][4 * q + 3] * unscale + b[4 * q + 3]);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cot0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * grp;
                if (ok && co <= cout1)
                    out[((size_t)n * Cout + co) * HW + p] = acc[t][r] * unscale + b[r];
            }
        }
    }
    if (!F32 && sat_count && __ballot(sat) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(sat_count, 1u);
}

// The pooled pixels the POOL epilogue left incomplete: first pooled row of every tile row but the top one (lacks the image row above the
// tile: the last row of the tile above), first pooled column of every tile column but the left one (lacks the column left of the tile).
// One work-item per such pixel and 8-channel group; lines = (tiles_y - 1) pooled rows of OW pixels, then (tiles_x - 1) pooled columns of OH.
__global__ __launch_bounds__(256) void pool_fix_kernel(float *__restrict__ out, const float *__restrict__ prow, const float *__restrict__ pcol,
                                                       int C8, int H, int W, int tiles_x, int tiles_y) {
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const int nrow = (tiles_y - 1) * OW, ncol = (tiles_x - 1) * OH;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrow + ncol) return;
    int py, px;
    if (idx < nrow) { py = (idx / OW + 1) * (CV_H / 2); px = idx % OW; }
    else {
        const int k = idx - nrow;
        px = (k / OH + 1) * (CV_W / 2); py = k % OH;
        if (py % (CV_H / 2) == 0 && py > 0) return;                 // a tile corner: done by its row line
    }
    if (py >= OH || px >= OW) return;
    const int n = blockIdx.y / C8, c8 = blockIdx.y % C8;
    const bool need_row = py % (CV_H / 2) == 0 && py > 0, need_col = px % (CV_W / 2) == 0 && px > 0;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto add = [&](const float *p) {
        const float4 u = reinterpret_cast<const float4 *>(p)[0], v = reinterpret_cast<const float4 *>(p)[1];
        s[0] += u.x; s[1] += u.y; s[2] += u.z; s[3] += u.w; s[4] += v.x; s[5] += v.y; s[6] += v.z; s[7] += v.w;
    };
    if (need_row) {
        const int ty = py / (CV_H / 2);
        const float *rb = prow + (((size_t)n * tiles_y + ty - 1) * C8 + c8) * (size_t)W * 8;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = 2 * px + dx;
            if (x >= 0 && x < W) add(rb + (size_t)x * 8);
        }
    }
    if (need_col) {
        const int tx = px / (CV_W / 2);
        const float *cb = pcol + (((size_t)n * tiles_x + tx - 1) * C8 + c8) * (size_t)H * 8;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = 2 * py + dy;
            if (dy < 0 && need_row) continue;                        // (that corner came with the row above)
            if (y >= 0 && y < H) add(cb + (size_t)y * 8);
        }
    }
    float4 *o = reinterpret_cast<float4 *>(out + ((((size_t)n * C8 + c8) * OH + py) * OW + px) * 8);
    float4 u = o[0], v = o[1];
    u.x += s[0] * (1.0f / 9.0f); u.y += s[1] * (1.0f / 9.0f); u.z += s[2] * (1.0f / 9.0f); u.w += s[3] * (1.0f / 9.0f);
    v.x += s[4] * (1.0f / 9.0f); v.y += s[5] * (1.0f / 9.0f); v.z += s[6] * (1.0f / 9.0f); v.w += s[7] * (1.0f / 9.0f);
    o[0] = u; o[1] = v;
}

// The border rows / columns of every 16 x 64 output block of the UPS epilogue (output rows 16t, 16t + 15, columns 64t, 64t + 63), from the side
// buffers alone: slr_upsample_bilinear2x's expression on the first / last rows and columns of the tiles.  Lines = 2 * tiles_y output rows of
// OW pixels, then 2 * tiles_x output columns of OH pixels (minus the pixels of the border rows).
__global__ __launch_bounds__(256) void upsample_fix_kernel(float *__restrict__ out, const float *__restrict__ prow, const float *__restrict__ pcol,
                                                           int C8, int H, int W, int tiles_x, int tiles_y) {
    const int OH = 2 * H, OW = 2 * W;
    const int nrow = 2 * tiles_y * OW, ncol = 2 * tiles_x * OH;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nrow + ncol) return;
    int R, C;
    bool rowline;
    if (idx < nrow) { const int l = idx / OW; R = (l >> 1) * 2 * CV_H + ((l & 1) ? 2 * CV_H - 1 : 0); C = idx % OW; rowline = true; }
    else {
        const int k = idx - nrow, l = k / OH;
        C = (l >> 1) * 2 * CV_W + ((l & 1) ? 2 * CV_W - 1 : 0); R = k % OH; rowline = false;
        const int rm = R % (2 * CV_H);
        if (rm == 0 || rm == 2 * CV_H - 1) return;                  // done by its row line
    }
    if (R >= OH || C >= OW) return;
    const int n = blockIdx.y / C8, c8 = blockIdx.y % C8;
    const float sy = fmaxf((R + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf((C + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int ya = (int)sy, xa = (int)sx;
    const float ly1 = sy - (float)ya, ly0 = 1.0f - ly1, lx1 = sx - (float)xa, lx0 = 1.0f - lx1;
    // the second row / column meets weight 0 at the image's first row / column (the two-kernel form reads row / column 1 there): same row / column
    const int yb = ly1 == 0.0f ? ya : min(ya + 1, H - 1), xb = lx1 == 0.0f ? xa : min(xa + 1, W - 1);
    auto get = [&](int y, int x, float (&v)[8]) {
        const float *p;
        if (rowline) p = prow + (((((size_t)n * tiles_y + y / CV_H) * 2 + (y % CV_H ? 1 : 0)) * C8 + c8) * W + x) * 8;
        else p = pcol + (((((size_t)n * tiles_x + x / CV_W) * 2 + (x % CV_W ? 1 : 0)) * C8 + c8) * H + y) * 8;
        const float4 u = reinterpret_cast<const float4 *>(p)[0], w = reinterpret_cast<const float4 *>(p)[1];
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; v[4] = w.x; v[5] = w.y; v[6] = w.z; v[7] = w.w;
    };
    float a00[8], a01[8], a10[8], a11[8], r[8];
    get(ya, xa, a00); get(ya, xb, a01); get(yb, xa, a10); get(yb, xb, a11);
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = ly0 * (lx0 * a00[c] + lx1 * a01[c]) + ly1 * (lx0 * a10[c] + lx1 * a11[c]);
    float4 *o = reinterpret_cast<float4 *>(out + ((((size_t)n * C8 + c8) * OH + R) * OW + C) * 8);
    o[0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1] = make_float4(r[4], r[5], r[6], r[7]);
}

// w [Cout,Cin,k,k] fp32 (taps = k*k = 9 or 1) -> split f16 weights in fragment order over the PADDED
// channel counts (zero weights for the padding), scaled by wscale (a power of two)
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ ws,
                                                                 int Cout, int Cin, int CoutP, int CinP, int taps,
                                                                 float wscale) {
    const int total = CoutP * CinP * taps;
    const int nchunk = CinP >> 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % taps, ci = (i / taps) % CinP, co = i / (taps * CinP);
        const float x = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] * wscale : 0.0f;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const size_t frag = (((size_t)(co >> 5) * nchunk + (ci >> 4)) * taps + tap) * 2;
        const int within = ((ci & 15) >> 3) * 256 + (co & 31) * 8 + (ci & 7);     // [ci group][co][8 ci]
        ws[frag * 512 + within] = h;
        ws[(frag + 1) * 512 + within] = l;
    }
}

// fp32 rung: w [Cout,Cin,k,k] fp32 -> fp32 weights in fragment order over the padded channel counts:
// [co tile][chunk of 16 ci][tap][lane 64][k pair 8], lane = (co & 31) + 32 * g, and the lane's k-th value is input channel
// chunk * 16 + (pair8 ? 8 * g + k : 2 * k + g): the 3x3 kernel pairs channels (2k, 2k + 1) (its LDS rows), the 1x1 kernel (k, 8 + k)
// (what its 8 plane loads per lane deliver).  Same bytes as the split-f16 buffer.
__global__ __launch_bounds__(256) void conv_f32_weights_kernel(const float *__restrict__ w, float *__restrict__ wf, int Cout, int Cin,
                                                               int CoutP, int CinP, int taps, int pair8) {
    const int total = CoutP * CinP * taps;
    const int nchunk = CinP >> 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % taps, ci = (i / taps) % CinP, co = i / (taps * CinP);
        const float x = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.0f;
        const int cl = ci & 15;
        const int g = pair8 ? cl >> 3 : cl & 1, k = pair8 ? cl & 7 : cl >> 1;
        const size_t frag = ((size_t)(co >> 5) * nchunk + (ci >> 4)) * taps + tap;
        wf[frag * 512 + ((co & 31) + 32 * g) * 8 + k] = x;
    }
}

// Saturation counter of the split-f16 kernels, ONE PER DEVICE (lazily allocated, zeroed; shared by every stream and host
// thread that runs convolutions on that device): the kernels add to it when an activation reaches the f16 range of the
// split (|x| * xscale >= 65472, i.e. |x| >= 1023 at the default 2^6) and had to be clamped.
static std::atomic<unsigned *> g_sat[64];
static std::mutex g_sat_lock;
static int sat_counter(unsigned **p) {
    int dev = 0;
    SLR_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { *p = nullptr; return 0; }
    if (!g_sat[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> guard(g_sat_lock);           // host threads may issue their first convolution together
        if (g_sat[dev].load(std::memory_order_relaxed)) { *p = g_sat[dev].load(); return 0; }
        unsigned *q = nullptr;
        hipError_t e = hipMalloc((void **)&q, 256);
        if (e != hipSuccess) {       // e.g. first convolution of a device inside a stream capture
            set_error("conv: cannot allocate the saturation counter (%s); call slr_conv_saturation_count once before "
                      "capturing a graph", hipGetErrorString(e));
            return (int)e;
        }
        SLR_CHECK_HIP(hipMemset(q, 0, 256));
        g_sat[dev].store(q, std::memory_order_release);
    }
    *p = g_sat[dev].load(std::memory_order_acquire);
    return 0;
}

}  // namespace slr
#ifdef SLR_TRACE
namespace slr { extern long long *g_trace; }       // slr_debug_trace (splat_op.hip)
#endif
#include "conv_wino.hpp"

using namespace slr;

SLR_EXPORT int slr_conv_saturation_count(unsigned long long *count, int reset, void *stream) {
    SLR_CHECK_ARG(count, "null pointer");
    unsigned *p = nullptr;
    if (int e = sat_counter(&p)) return e;
    unsigned v = 0;
    if (p) {
        // ordered on the CALLER's stream (the legacy null stream does not wait for hipStreamNonBlocking streams, which is
        // what torch's side streams are): the convolutions enqueued on `stream` before this call are counted
        hipStream_t st = (hipStream_t)stream;
        SLR_CHECK_HIP(hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, st));
        SLR_CHECK_HIP(hipStreamSynchronize(st));
        if (reset && v) { SLR_CHECK_HIP(hipMemsetAsync(p, 0, sizeof(v), st)); SLR_CHECK_HIP(hipStreamSynchronize(st)); }
    }
    *count = v;
    return 0;
}

SLR_EXPORT int slr_conv_saturation_record(unsigned *host_slot, void *stream) {
    SLR_CHECK_ARG(host_slot, "null pointer");
    unsigned *p = nullptr;
    if (int e = sat_counter(&p)) return e;
    if (!p) { *host_slot = 0; return 0; }
    SLR_CHECK_HIP(hipMemcpyAsync(host_slot, p, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

// Cout <= 4: the weight buffer holds plain fp32 weights [ci padded to 8][tap][4] on either rung (conv_few.hpp); it fits the split layout's bytes
static int conv_few_weights(const float *w, void *wbuf, int Cout, int Cin, hipStream_t st) {
    const int CinP = conv_few_cin_pad(Cin);
    static_assert(36 * sizeof(float) * 8 <= 32 * 16 * 9 * 2 * sizeof(_Float16), "8 input channels of plain weights fit 16 of the split layout");
    hipLaunchKernelGGL(conv_few_weights_kernel, dim3((CinP * 36 + 255) / 256), dim3(256), 0, st, w, (float *)wbuf, Cout, Cin, CinP);
    SLR_CHECK_LAUNCH();
    return 0;
}

template <int NCO>
static int conv_few_launch(ConvArgs &a, bool in_b8, hipStream_t st) {
    const dim3 grid(((a.W + CF_BW - 1) / CF_BW) * ((a.H + CF_BH - 1) / CF_BH), 1, a.N);
    if (a.skip_out) {                                   // (the *_skipout entry points: channel-blocked input)
        if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_few_kernel<NCO, true, true, true>), grid, dim3(CF_THREADS), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_few_kernel<NCO, false, true, true>), grid, dim3(CF_THREADS), 0, st, a);
        SLR_CHECK_LAUNCH();
        return 0;
    }
    if (a.pre != PRE_NONE && in_b8) hipLaunchKernelGGL((conv3x3_few_kernel<NCO, true, true>), grid, dim3(CF_THREADS), 0, st, a);
    else if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_few_kernel<NCO, true, false>), grid, dim3(CF_THREADS), 0, st, a);
    else if (in_b8) hipLaunchKernelGGL((conv3x3_few_kernel<NCO, false, true>), grid, dim3(CF_THREADS), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_few_kernel<NCO, false, false>), grid, dim3(CF_THREADS), 0, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT size_t slr_conv3x3_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)conv_cout_pad(Cout) * conv_cin_pad(Cin) * 9 * 2 * sizeof(_Float16);
}

SLR_EXPORT int slr_conv3x3_split_weights(const float *w, void *wsplit, int Cout, int Cin, float wscale, void *stream) {
    SLR_CHECK_ARG(w && wsplit, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)conv_cout_pad(Cout) * conv_cin_pad(Cin) * 9 < (1LL << 30), "sizes");
    SLR_CHECK_ARG(wscale > 0.0f, "wscale");
    if (Cout <= CF_MAXCO) return conv_few_weights(w, wsplit, Cout, Cin, (hipStream_t)stream);      // (plain fp32 weights: conv_few.hpp)
    const int CoutP = conv_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    const int total = CoutP * CinP * 9;
    hipLaunchKernelGGL(conv_split_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16 *)wsplit, Cout, Cin, CoutP, CinP, 9, wscale);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int conv_check_layout(int layout, const void *in, const void *out, int Cin, int Cout, const void *residual,
                             bool derived_mask);

// 1x1: output channels in groups of NCT*32 <= 128 per workgroup row (64 accumulator registers: 3 waves per SIMD;
// wider layers re-read the input once per 128 channels, mostly from L2; 64-channel rows measured the same)
static int conv1x1_nct(int Cout) { const int t = (Cout + 31) / 32; return t > 2 ? 4 : (t > 1 ? 2 : 1); }
static int conv1x1_cout_pad(int Cout) { const int g = conv1x1_nct(Cout) * 32; return (Cout + g - 1) / g * g; }

SLR_EXPORT size_t slr_conv1x1_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)conv1x1_cout_pad(Cout) * conv_cin_pad(Cin) * 2 * sizeof(_Float16);
}

SLR_EXPORT int slr_conv1x1_split_weights(const float *w, void *wsplit, int Cout, int Cin, float wscale, void *stream) {
    SLR_CHECK_ARG(w && wsplit, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)conv1x1_cout_pad(Cout) * conv_cin_pad(Cin) < (1LL << 30), "sizes");
    SLR_CHECK_ARG(wscale > 0.0f, "wscale");
    const int CoutP = conv1x1_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    const int total = CoutP * CinP;
    hipLaunchKernelGGL(conv_split_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16 *)wsplit, Cout, Cin, CoutP, CinP, 1, wscale);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_conv3x3_f32_weights(const float *w, void *wfrag, int Cout, int Cin, void *stream) {
    SLR_CHECK_ARG(w && wfrag, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)conv_cout_pad(Cout) * conv_cin_pad(Cin) * 9 < (1LL << 30), "sizes");
    if (Cout <= CF_MAXCO) return conv_few_weights(w, wfrag, Cout, Cin, (hipStream_t)stream);
    const int CoutP = conv_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    const int total = CoutP * CinP * 9;
    hipLaunchKernelGGL(conv_f32_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (float *)wfrag,
                       Cout, Cin, CoutP, CinP, 9, 0);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT size_t slr_conv3x3_wino_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)wino_cout_pad(Cout) * conv_cin_pad(Cin) * 16 * sizeof(float);
}

SLR_EXPORT int slr_conv3x3_wino_weights(const float *w, void *wfrag, int Cout, int Cin, void *stream) {
    SLR_CHECK_ARG(w && wfrag, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)wino_cout_pad(Cout) * conv_cin_pad(Cin) * 16 < (1LL << 30), "sizes");
    const int CoutP = wino_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    hipLaunchKernelGGL(conv_wino_weights_kernel, dim3((CoutP * CinP + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (float *)wfrag,
                       Cout, Cin, CoutP, CinP);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_conv1x1_f32_weights(const float *w, void *wfrag, int Cout, int Cin, void *stream) {
    SLR_CHECK_ARG(w && wfrag, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)conv1x1_cout_pad(Cout) * conv_cin_pad(Cin) < (1LL << 30), "sizes");
    const int CoutP = conv1x1_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    const int total = CoutP * CinP;
    hipLaunchKernelGGL(conv_f32_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (float *)wfrag,
                       Cout, Cin, CoutP, CinP, 1, 1);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int check_xscale(float xscale) {
    int ex = 0;
    SLR_CHECK_ARG(xscale > 0.0f && xscale <= CV_XSCALE && frexpf(xscale, &ex) == 0.5f, "xscale: a power of two in (0, 64]");
    return 0;
}

SLR_EXPORT int slr_conv1x1_forward(const float *in, const void *wsplit, const float *bias, float *out, int N, int Cin,
                                   int Cout, int H, int W, float wscale, float xscale, int layout, void *stream) {
    SLR_CHECK_ARG(in && wsplit && out, "null pointer");
    if (int e = check_xscale(xscale)) return e;
    const bool f32 = (layout & SLR_CONV_F32) != 0;
    SLR_CHECK_ARG(!f32 || (wscale == 1.0f && xscale == 1.0f), "the fp32 rung takes no operand scales (wscale = xscale = 1)");
    layout &= ~SLR_CONV_F32;
    if (int e = conv_check_layout(layout & ~SLR_CONV_RES_B8, in, out, Cin, Cout, nullptr, false)) return e;
    SLR_CHECK_ARG(!(layout & SLR_CONV_RES_B8), "layout flags");
    SLR_CHECK_ARG(N > 0 && N < 65536 && Cin > 0 && Cout > 0 && Cout < (1 << 20) && H > 0 && W > 0 &&
                  (long long)Cin * H * W < (1LL << 40) && (long long)H * W < (1LL << 31) - 128, "sizes");
    const int HW = H * W, nchunk = conv_cin_pad(Cin) / 16, nct = conv1x1_nct(Cout);
    const float unscale = 1.0f / (xscale * wscale);
    const dim3 grid((HW + 128 * C1_TILES - 1) / (128 * C1_TILES), conv1x1_cout_pad(Cout) / (nct * 32), N);
    hipStream_t st = (hipStream_t)stream;
    const int ob8 = (layout & SLR_CONV_OUT_B8) ? 1 : 0;
    unsigned *satp = nullptr;
    if (int e = sat_counter(&satp)) return e;
#define C1_LAUNCH(T)                                                                                                       \
    do {                                                                                                                   \
        if (f32 && (layout & SLR_CONV_IN_B8)) hipLaunchKernelGGL((conv1x1_split_kernel<T, true, true>), grid, dim3(256), 0, st, in, (const h8 *)wsplit, bias, out, Cin, Cout, HW, nchunk, unscale, xscale, ob8, satp); \
        else if (f32) hipLaunchKernelGGL((conv1x1_split_kernel<T, false, true>), grid, dim3(256), 0, st, in, (const h8 *)wsplit, bias, out, Cin, Cout, HW, nchunk, unscale, xscale, ob8, satp); \
        else if (layout & SLR_CONV_IN_B8) hipLaunchKernelGGL((conv1x1_split_kernel<T, true>), grid, dim3(256), 0, st, in, (const h8 *)wsplit, bias, out, Cin, Cout, HW, nchunk, unscale, xscale, ob8, satp); \
        else hipLaunchKernelGGL((conv1x1_split_kernel<T, false>), grid, dim3(256), 0, st, in, (const h8 *)wsplit, bias, out, Cin, Cout, HW, nchunk, unscale, xscale, ob8, satp); \
    } while (0)
    if (nct == 4) C1_LAUNCH(4); else if (nct == 2) C1_LAUNCH(2); else C1_LAUNCH(1);
#undef C1_LAUNCH
    SLR_CHECK_LAUNCH();
    return 0;
}

template <bool F32>
static int conv_launch_t(ConvArgs &a, bool in_b8, hipStream_t st);

static int conv_wino_launch(ConvArgs &a, bool in_b8, hipStream_t st) {
    static LdsOptIn attr[4];
    const int which = (a.pre != PRE_NONE ? 2 : 0) + (in_b8 ? 1 : 0);
    const void *fn = which == 3 ? (const void *)conv3x3_wino_kernel<true, true> : which == 2 ? (const void *)conv3x3_wino_kernel<true, false>
                   : which == 1 ? (const void *)conv3x3_wino_kernel<false, true> : (const void *)conv3x3_wino_kernel<false, false>;
    if (int e = lds_opt_in(fn, (int)WN_LDS_BYTES, attr[which])) return e;
    a.tiles_x = (a.W + WN_BW - 1) / WN_BW;
    a.nchunk = conv_cin_pad(a.Cin) / 16;
    a.xscale = 1.0f; a.unscale = 1.0f;
    const int blocks = a.tiles_x * ((a.H + WN_BH - 1) / WN_BH), ngrp = wino_cout_pad(a.Cout) / 64;
    const dim3 grid((blocks + WN_SLICE - 1) / WN_SLICE * WN_SLICE * ngrp, 1, a.N);
    a.wino_groups = ngrp;
#ifdef SLR_TRACE
    a.trace = g_trace;
#endif
    if (which == 3) hipLaunchKernelGGL((conv3x3_wino_kernel<true, true>), grid, dim3(WN_THREADS), WN_LDS_BYTES, st, a);
    else if (which == 2) hipLaunchKernelGGL((conv3x3_wino_kernel<true, false>), grid, dim3(WN_THREADS), WN_LDS_BYTES, st, a);
    else if (which == 1) hipLaunchKernelGGL((conv3x3_wino_kernel<false, true>), grid, dim3(WN_THREADS), WN_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv3x3_wino_kernel<false, false>), grid, dim3(WN_THREADS), WN_LDS_BYTES, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int conv_launch(ConvArgs &a, float wscale, float xscale, bool in_b8, bool f32, hipStream_t st, bool wino = false) {
    if (int e = check_xscale(xscale)) return e;
    SLR_CHECK_ARG(!f32 || (wscale == 1.0f && xscale == 1.0f), "the fp32 rung takes no operand scales (wscale = xscale = 1)");
    SLR_CHECK_ARG(!wino || (f32 && a.Cout > CF_MAXCO), "SLR_CONV_WINO goes with SLR_CONV_F32 and more than 4 output channels");
    SLR_CHECK_ARG(!wino || a.pre == PRE_NONE || a.Cin <= WN_MAXCIN, "SLR_CONV_WINO with a prologue supports Cin <= 256");
    if (wino) return conv_wino_launch(a, in_b8, st);
    if (a.Cout <= CF_MAXCO) {                           // fp32 FMAs on the vector ALUs on either rung: no operand scales, nothing saturates
        switch (a.Cout) {
            case 1: return conv_few_launch<1>(a, in_b8, st);
            case 2: return conv_few_launch<2>(a, in_b8, st);
            case 3: return conv_few_launch<3>(a, in_b8, st);
            default: return conv_few_launch<4>(a, in_b8, st);
        }
    }
    a.tiles_x = (a.W + CV_W - 1) / CV_W;
    a.nchunk = conv_cin_pad(a.Cin) / 16;
    a.xscale = xscale;
    a.unscale = 1.0f / (xscale * wscale);
    if (int e = sat_counter(&a.sat)) return e;
    return f32 ? conv_launch_t<true>(a, in_b8, st) : conv_launch_t<false>(a, in_b8, st);
}

template <bool F32>
static int conv_launch_t(ConvArgs &a, bool in_b8, hipStream_t st) {
    const int tiles = a.tiles_x * ((a.H + CV_H - 1) / CV_H);
    int ct = conv_cout_tile(a.Cout);
    // NCHW input WITH a prologue and > 64 output channels: 24 scalar staging loads + the prologue table + 128
    // accumulator registers do not fit 256 VGPRs (the <1,4,true,false> instantiation spilled 14-27 of them); that
    // combination runs as two 64-channel workgroup rows instead (no network on the path uses it: wide layers read
    // channel-blocked activations).  The weight buffer is indexed by 32-channel tiles, so any row width reads it.
    if (ct == 128 && a.pre != PRE_NONE && !in_b8) ct = 64;
    const dim3 grid(tiles, conv_cout_pad(a.Cout) / ct, a.N);
    {
        if (a.skip_in) {                    // (conv_set_skip: channel-blocked main input, more than 4 output channels; fp32 rung: more than 64)
#define CV_SKIP(WCO)                                                                                            \
    do {                                                                                                       \
        if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_split_kernel<1, WCO, true, true, F32, true>), grid, dim3(CV_THREADS), 0, st, a);   \
        else hipLaunchKernelGGL((conv3x3_split_kernel<1, WCO, false, true, F32, true>), grid, dim3(CV_THREADS), 0, st, a);                   \
    } while (0)
            if (a.resample) {               // (conv_set_skip: 128-channel workgroup rows)
                if (a.resample == 2) {
                    if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, true, true, F32, true, false, true>), grid, dim3(CV_THREADS), 0, st, a);
                    else hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, false, true, F32, true, false, true>), grid, dim3(CV_THREADS), 0, st, a);
                    SLR_CHECK_LAUNCH();
                    const int items = 2 * a.tiles_y * 2 * a.W + 2 * a.tiles_x * 2 * a.H;
                    hipLaunchKernelGGL(upsample_fix_kernel, dim3((items + 255) / 256, a.N * (a.Cout >> 3)), dim3(256), 0, st, a.out, (const float *)a.pool_row,
                                       (const float *)a.pool_col, a.Cout >> 3, a.H, a.W, a.tiles_x, a.tiles_y);
                    SLR_CHECK_LAUNCH();
                    return 0;
                }
                if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, true, true, F32, true, true>), grid, dim3(CV_THREADS), 0, st, a);
                else hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, false, true, F32, true, true>), grid, dim3(CV_THREADS), 0, st, a);
                SLR_CHECK_LAUNCH();
                const int OH = (a.H - 1) / 2 + 1, OW = (a.W - 1) / 2 + 1;
                const int items = (a.tiles_y - 1) * OW + (a.tiles_x - 1) * OH;
                if (items > 0) {
                    hipLaunchKernelGGL(pool_fix_kernel, dim3((items + 255) / 256, a.N * (a.Cout >> 3)), dim3(256), 0, st, a.out, (const float *)a.pool_row,
                                       (const float *)a.pool_col, a.Cout >> 3, a.H, a.W, a.tiles_x, a.tiles_y);
                    SLR_CHECK_LAUNCH();
                }
                return 0;
            }
            if (ct == 128) CV_SKIP(4);
            else if constexpr (!F32) { if (ct == 64) CV_SKIP(2); else CV_SKIP(1); }
#undef CV_SKIP
            SLR_CHECK_LAUNCH();
            return 0;
        }
    }
#define CV_LAUNCH(CPW, WCO)                                                                                     \
    do {                                                                                                       \
        if (a.pre != PRE_NONE && in_b8) hipLaunchKernelGGL((conv3x3_split_kernel<CPW, WCO, true, true, F32>), grid, dim3(CV_THREADS), 0, st, a);    \
        else if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_split_kernel<CPW, WCO, true, false, F32>), grid, dim3(CV_THREADS), 0, st, a);      \
        else if (in_b8) hipLaunchKernelGGL((conv3x3_split_kernel<CPW, WCO, false, true, F32>), grid, dim3(CV_THREADS), 0, st, a);                  \
        else hipLaunchKernelGGL((conv3x3_split_kernel<CPW, WCO, false, false, F32>), grid, dim3(CV_THREADS), 0, st, a);                            \
    } while (0)
    if (ct == 128) {                        // one 32-channel tile x all 8 rows per wave: a quarter of the weight-fragment
                                            // traffic of 4 x 2 tiles per wave would need, half of <2,2> (+3..5 % measured)
        if (a.pre != PRE_NONE) hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, true, true, F32>), grid, dim3(CV_THREADS), 0, st, a);   // (in_b8)
        else if (in_b8) hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, false, true, F32>), grid, dim3(CV_THREADS), 0, st, a);
        else hipLaunchKernelGGL((conv3x3_split_kernel<1, 4, false, false, F32>), grid, dim3(CV_THREADS), 0, st, a);
    }
    else if (ct == 64) CV_LAUNCH(1, 2);     // 1 tile x 4 rows per wave: half the weight-fragment loads of <2,1> (+4 %)
    else CV_LAUNCH(1, 1);
#undef CV_LAUNCH
    SLR_CHECK_LAUNCH();
    return 0;
}

static int conv_check_layout(int layout, const void *in, const void *out, int Cin, int Cout, const void *residual,
                             bool derived_mask) {
    layout &= ~(SLR_CONV_F32 | SLR_CONV_WINO | SLR_CONV_SKIP_B8 | SLR_CONV_POOL_OUT | SLR_CONV_UP_OUT);      // (the last three: conv_set_skip)
    SLR_CHECK_ARG((layout & ~(SLR_CONV_IN_B8 | SLR_CONV_OUT_B8 | SLR_CONV_RES_B8)) == 0, "layout flags");
    SLR_CHECK_ARG(!(layout & SLR_CONV_RES_B8) || ((layout & SLR_CONV_OUT_B8) && residual && !((uintptr_t)residual & 15)),
                  "a channel-blocked residual goes with a channel-blocked output");
    SLR_CHECK_ARG(!(layout & SLR_CONV_IN_B8) || (Cin % 8 == 0 && !((uintptr_t)in & 15) && !derived_mask),
                  "channel-blocked input needs Cin % 8 == 0, a 16-byte aligned tensor and an explicit mask");
    SLR_CHECK_ARG(!(layout & SLR_CONV_OUT_B8) || (Cout % 8 == 0 && !((uintptr_t)out & 15)),
                  "channel-blocked output needs Cout % 8 == 0 and a 16-byte aligned tensor");
    return 0;
}

static int conv_check_dims(int N, int Cin, int Cout, int H, int W) {
    SLR_CHECK_ARG(N > 0 && N < 65536 && Cin > 0 && Cout > 0 && Cout < (1 << 20) && H > 0 && W > 0 &&
                  (long long)Cin * H * W < (1LL << 31) && (long long)N * Cout * H * W < (1LL << 40), "sizes");
    return 0;
}

// The 1x1 skip branch riding in the 3x3 kernel (slr_conv3x3_forward_skip / slr_pconv3x3_forward_skip)
struct SkipOp { const float *in; const void *w; const float *bias; int cin; float wscale; void *pool_ws; size_t pool_ws_bytes;
                float *skip_out; };        // skip_out: the *_skipout form (Cout <= 4: the skip of the SAME input written next to the result; w = [Cin][4] fp32)

// side buffers of the pooling epilogue: the last row of every tile row and the last column of every tile column of the result
static size_t conv_pool_ws_bytes(int N, int Cout, int H, int W, bool up = false) {
    const size_t tiles_x = (W + CV_W - 1) / CV_W, tiles_y = (H + CV_H - 1) / CV_H;
    return (size_t)N * Cout * (tiles_y * W + tiles_x * H) * sizeof(float) * (up ? 2 : 1);       // (up-sampling: first AND last row / column)
}

SLR_EXPORT size_t slr_conv_pool_ws_bytes(int N, int Cout, int H, int W) {
    if (N <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    return conv_pool_ws_bytes(N, Cout, H, W);
}

SLR_EXPORT size_t slr_conv_up_ws_bytes(int N, int Cout, int H, int W) {
    if (N <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    return conv_pool_ws_bytes(N, Cout, H, W, true);
}

static int conv_set_skip(ConvArgs &a, const SkipOp *sk, float xscale, int &layout) {
    const int layout_in = layout;
    const bool sb8 = (layout & SLR_CONV_SKIP_B8) != 0, up = (layout & SLR_CONV_UP_OUT) != 0, pool = (layout & SLR_CONV_POOL_OUT) != 0 || up;
    layout &= ~(SLR_CONV_SKIP_B8 | SLR_CONV_POOL_OUT | SLR_CONV_UP_OUT);
    if (!sk) { SLR_CHECK_ARG(!sb8 && !pool, "layout flags"); return 0; }
    if (sk->skip_out) {
        SLR_CHECK_ARG(!sb8 && !pool, "layout flags");
        SLR_CHECK_ARG(sk->w && a.Cout <= CF_MAXCO && (layout & SLR_CONV_IN_B8) && !((uintptr_t)sk->w & 15),
                      "the skip output needs Cout <= 4, a channel-blocked input and 16-byte aligned [Cin][4] weights");
        SLR_CHECK_ARG(a.pre != PRE_BN_NONZERO, "the skip output goes with an explicit mask");
        a.skip_w = (const h8 *)sk->w; a.skip_bias = sk->bias; a.skip_out = sk->skip_out;
        return 0;
    }
    SLR_CHECK_ARG(sk->in && sk->w, "null pointer");
    SLR_CHECK_ARG(!(layout & SLR_CONV_WINO), "no fused skip branch in the Winograd kernel");
    SLR_CHECK_ARG(!(layout & SLR_CONV_F32) || (conv_cout_tile(a.Cout) == 128 && sk->wscale == 1.0f),
                  "the fused skip branch on the fp32 rung: more than 64 output channels, skip_wscale = 1");
    SLR_CHECK_ARG((layout & SLR_CONV_IN_B8) && a.Cout > CF_MAXCO, "the fused skip branch needs a channel-blocked main input and more than 4 output channels");
    SLR_CHECK_ARG(sk->cin > 0 && (long long)sk->cin * a.H * a.W < (1LL << 31) && sk->wscale > 0.0f, "skip sizes");
    SLR_CHECK_ARG(sb8 && sk->cin % 8 == 0 && !((uintptr_t)sk->in & 15), "the fused skip branch reads a channel-blocked skip input (SLR_CONV_SKIP_B8): skip_cin % 8 == 0, 16-byte aligned");
    SLR_CHECK_ARG(!a.residual && !a.next_scale, "the fused skip branch replaces the residual; no next-BN fusion with it");
    a.skip_in = sk->in; a.skip_w = (const h8 *)sk->w; a.skip_bias = sk->bias;
    a.skip_cin = sk->cin; a.skip_nchunk = conv_cin_pad(sk->cin) / 16;
    a.skip_unscale = 1.0f / (xscale * sk->wscale);
    if (pool) {
        SLR_CHECK_ARG(!(up && (layout_in & SLR_CONV_POOL_OUT)), "SLR_CONV_POOL_OUT and SLR_CONV_UP_OUT are exclusive");
        SLR_CHECK_ARG(a.out_b8 && conv_cout_tile(a.Cout) == 128, "the resampling epilogues need a channel-blocked output and more than 64 output channels");
        SLR_CHECK_ARG(!up || (long long)a.N * a.Cout * a.H * a.W < (1LL << 38), "sizes");
        SLR_CHECK_ARG(sk->pool_ws && !((uintptr_t)sk->pool_ws & 15) && sk->pool_ws_bytes >= conv_pool_ws_bytes(a.N, a.Cout, a.H, a.W, up),
                      "pool_ws: slr_conv_pool_ws_bytes / slr_conv_up_ws_bytes, 16-byte aligned");
        a.tiles_y = (a.H + CV_H - 1) / CV_H;
        a.resample = up ? 2 : 1;
        a.pool_row = (float *)sk->pool_ws;
        a.pool_col = a.pool_row + (size_t)a.N * a.Cout * a.tiles_y * a.W * (up ? 2 : 1);
    }
    return 0;
}

static int conv3x3_forward_impl(const float *in, const void *wsplit, const float *bias, const float *residual,
                                   float *out, int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                                   const float *pre_scale, const float *pre_shift, const SkipOp *sk, int layout, void *stream) {
    SLR_CHECK_ARG(in && wsplit && out, "null pointer");
    if (int e = conv_check_layout(layout, in, out, Cin, Cout, residual, false)) return e;
    SLR_CHECK_ARG(!pre_scale == !pre_shift, "pre_scale / pre_shift go together");
    SLR_CHECK_ARG(!pre_scale || Cin <= CV_MAXCIN, "prologue supports Cin <= 1024");
    if (int e = conv_check_dims(N, Cin, Cout, H, W)) return e;
    ConvArgs a = {};
    a.in = in; a.w = (const h8 *)wsplit; a.bias = bias; a.out = out;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.pre = pre_scale ? PRE_BN : PRE_NONE;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift;
    a.residual = residual;
    a.out_b8 = (layout & SLR_CONV_OUT_B8) != 0;
    a.res_b8 = (layout & SLR_CONV_RES_B8) != 0;
    if (int e = conv_set_skip(a, sk, xscale, layout)) return e;
    return conv_launch(a, wscale, xscale, (layout & SLR_CONV_IN_B8) != 0, (layout & SLR_CONV_F32) != 0, (hipStream_t)stream, (layout & SLR_CONV_WINO) != 0);
}

SLR_EXPORT int slr_conv3x3_forward(const float *in, const void *wsplit, const float *bias, const float *residual,
                                   float *out, int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                                   const float *pre_scale, const float *pre_shift, int layout, void *stream) {
    return conv3x3_forward_impl(in, wsplit, bias, residual, out, N, Cin, Cout, H, W, wscale, xscale, pre_scale, pre_shift, nullptr, layout, stream);
}

SLR_EXPORT int slr_conv3x3_forward_skip(const float *in, const void *wsplit, const float *bias, float *out,
                                        int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                                        const float *pre_scale, const float *pre_shift,
                                        const float *skip_in, const void *skip_wsplit, const float *skip_bias, int skip_cin, float skip_wscale,
                                        void *pool_ws, size_t pool_ws_bytes, int layout, void *stream) {
    const SkipOp sk = {skip_in, skip_wsplit, skip_bias, skip_cin, skip_wscale, pool_ws, pool_ws_bytes, nullptr};
    return conv3x3_forward_impl(in, wsplit, bias, nullptr, out, N, Cin, Cout, H, W, wscale, xscale, pre_scale, pre_shift, &sk, layout, stream);
}

SLR_EXPORT int slr_conv3x3_forward_skipout(const float *in, const void *wsplit, const float *bias, const float *residual, float *out,
                                           int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                                           const float *pre_scale, const float *pre_shift,
                                           const float *skip_w4, const float *skip_bias, float *skip_out, int layout, void *stream) {
    SLR_CHECK_ARG(skip_w4 && skip_out, "null pointer");
    const SkipOp sk = {nullptr, skip_w4, skip_bias, 0, 1.0f, nullptr, 0, skip_out};
    return conv3x3_forward_impl(in, wsplit, bias, residual, out, N, Cin, Cout, H, W, wscale, xscale, pre_scale, pre_shift, &sk, layout, stream);
}

static int pconv3x3_forward_impl(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                                    const void *wsplit, float wscale, float xscale, const float *bias, const float *residual,
                                    const float *next_scale, const float *next_shift, float *out, float *um_out,
                                    int N, int Cin, int Cout, int H, int W, const SkipOp *sk, int layout, void *stream) {
    SLR_CHECK_ARG(x && wsplit && bias && out, "null pointer");
    if (int e = conv_check_layout(layout, x, out, Cin, Cout, residual, mask == nullptr)) return e;
    SLR_CHECK_ARG(!pre_scale == !pre_shift, "pre_scale / pre_shift go together");
    SLR_CHECK_ARG(!pre_scale || Cin <= CV_MAXCIN, "prologue supports Cin <= 1024");
    SLR_CHECK_ARG(mask || pre_scale, "mask = NULL (derived from x != 0) needs the raw input, i.e. pre_scale / pre_shift");
    SLR_CHECK_ARG(!next_scale == !next_shift, "next_scale / next_shift go together");
    SLR_CHECK_ARG(!(residual && next_scale), "residual and next-BN fusion are exclusive");
    if (int e = conv_check_dims(N, Cin, Cout, H, W)) return e;
    ConvArgs a = {};
    a.in = x; a.w = (const h8 *)wsplit; a.bias = bias; a.out = out;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.pre = !pre_scale ? PRE_NONE : (mask ? PRE_BN_MASK : PRE_BN_NONZERO);
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.mask = mask;
    a.partial = 1;
    a.mask_scale = mask ? (float)Cin : 1.0f;           // channel-uniform mask: Cin identical planes (partialconv2d.py:61)
    a.winsize = (float)Cin * 9.0f;
    a.residual = residual; a.next_scale = next_scale; a.next_shift = next_shift; a.um_out = um_out;
    a.out_b8 = (layout & SLR_CONV_OUT_B8) != 0;
    a.res_b8 = (layout & SLR_CONV_RES_B8) != 0;
    if (int e = conv_set_skip(a, sk, xscale, layout)) return e;
    return conv_launch(a, wscale, xscale, (layout & SLR_CONV_IN_B8) != 0, (layout & SLR_CONV_F32) != 0, (hipStream_t)stream, (layout & SLR_CONV_WINO) != 0);
}

SLR_EXPORT int slr_pconv3x3_forward(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                                    const void *wsplit, float wscale, float xscale, const float *bias, const float *residual,
                                    const float *next_scale, const float *next_shift, float *out, float *um_out,
                                    int N, int Cin, int Cout, int H, int W, int layout, void *stream) {
    return pconv3x3_forward_impl(x, pre_scale, pre_shift, mask, wsplit, wscale, xscale, bias, residual, next_scale, next_shift, out, um_out,
                                 N, Cin, Cout, H, W, nullptr, layout, stream);
}

SLR_EXPORT int slr_pconv3x3_forward_skip(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                                         const void *wsplit, float wscale, float xscale, const float *bias, float *out, float *um_out,
                                         int N, int Cin, int Cout, int H, int W,
                                         const float *skip_in, const void *skip_wsplit, int skip_cin, float skip_wscale,
                                         void *pool_ws, size_t pool_ws_bytes, int layout, void *stream) {
    const SkipOp sk = {skip_in, skip_wsplit, nullptr, skip_cin, skip_wscale, pool_ws, pool_ws_bytes, nullptr};
    return pconv3x3_forward_impl(x, pre_scale, pre_shift, mask, wsplit, wscale, xscale, bias, nullptr, nullptr, nullptr, out, um_out,
                                 N, Cin, Cout, H, W, &sk, layout, stream);
}

SLR_EXPORT int slr_pconv3x3_forward_skipout(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                                            const void *wsplit, float wscale, float xscale, const float *bias, const float *residual,
                                            const float *next_scale, const float *next_shift, float *out, float *um_out,
                                            int N, int Cin, int Cout, int H, int W,
                                            const float *skip_w4, float *skip_out, int layout, void *stream) {
    SLR_CHECK_ARG(skip_w4 && skip_out, "null pointer");
    const SkipOp sk = {nullptr, skip_w4, nullptr, 0, 1.0f, nullptr, 0, skip_out};
    return pconv3x3_forward_impl(x, pre_scale, pre_shift, mask, wsplit, wscale, xscale, bias, residual, next_scale, next_shift, out, um_out,
                                 N, Cin, Cout, H, W, &sk, layout, stream);
}
