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
// fp32 Winograd kernels (1-6e-6), at 3x their speed (tools/convbench.py).
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
//   * wave (wc, wp) owns CPW 32-channel tiles x PT rows: CPW x PT accumulator tiles of 32x32
//     (128 registers for the 128-channel variant), 3 MFMAs per tile, tap and chunk;
//   * epilogue on the accumulators: plain (+bias) or the partial-convolution one
//     (o = (raw*ratio + b)*um, then + residual or the next layer's relu(bn(.))*um), same operations
//     in the same order as csrc/pconv.hip.  The D layout (row = channel, column = pixel) makes every
//     accumulator register a pair of 128-byte row segments of the NCHW output.
#include "slr_common.hpp"

namespace slr {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CV_W = 32, CV_H = 8;                 // output block
constexpr int CV_HW = CV_W + 2, CV_HH = CV_H + 2;  // input halo block
constexpr int CV_NPX = CV_HW * CV_HH;              // 340 halo pixels
constexpr int CV_ITEMS = 2 * CV_NPX;               // (pixel, 8-channel group) staging items per chunk
constexpr int CV_THREADS = 256;
constexpr int CV_ROUNDS = (CV_ITEMS + CV_THREADS - 1) / CV_THREADS;   // 3
constexpr float CV_XSCALE = 64.0f;                 // activations are scaled by 2^6 before the split
constexpr int CV_MAXCIN = 1024;                    // prologue scale/shift table in LDS

enum { PRE_NONE = 0, PRE_BN = 1, PRE_BN_MASK = 2, PRE_BN_NONZERO = 3 };

struct ConvArgs {
    const float *in;       // [N,Cin,H,W]
    const h8 *w;           // split weights in fragment order (see above), scaled by wscale
    float *out;            // [N,Cout,H,W]
    int N, Cin, Cout, H, W, tiles_x, nchunk;
    float unscale;         // 1 / (CV_XSCALE * wscale)
    // prologue: relu(x*pre_scale[c] - pre_shift[c]) * mask
    int pre;
    const float *pre_scale, *pre_shift, *pre_mask;     // [Cin], [Cin], [N,1,H,W]
    // epilogue
    const float *bias;     // [Cout] or nullptr
    int partial;           // partial-convolution epilogue (bias required)
    const float *mask_box; // [N,1,H,W] k x k box sum of the mask plane
    float mask_scale, winsize;
    const float *residual, *next_scale, *next_shift;
    float *um_out;         // [N,1,H,W] or nullptr
};

// padded channel counts of the weight buffer (shared by the split and the forward entry points)
__host__ __device__ inline int conv_cout_tile(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }
__host__ __device__ inline int conv_cout_pad(int Cout) { const int t = conv_cout_tile(Cout); return (Cout + t - 1) / t * t; }
__host__ __device__ inline int conv_cin_pad(int Cin) { return (Cin + 15) / 16 * 16; }

// CPW: 32-channel output tiles per wave; WCO: waves along the output channels (workgroup covers
// 32*CPW*WCO channels); the other 4/WCO wave rows split the 8 block rows.
template <int CPW, int WCO>
__global__ __launch_bounds__(CV_THREADS, 2) void conv3x3_split_kernel(ConvArgs a) {
    constexpr int WPX = 4 / WCO, PT = CV_H / WPX;
    __shared__ h8 xs[2][2][2][CV_NPX];             // [buffer][hi|lo][8-channel group][halo pixel]
    __shared__ __attribute__((aligned(16))) float pss[2][CV_MAXCIN];   // prologue scale / shift per (padded) input channel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    if (pre != PRE_NONE) {
        for (int i = tid; i < nchunk * 16; i += CV_THREADS) {
            pss[0][i] = i < a.Cin ? a.pre_scale[i] : 0.0f;
            pss[1][i] = i < a.Cin ? a.pre_shift[i] : 0.0f;
        }
    }

    // staging items of this work-item: (halo pixel, 8-channel group), constant over the chunks
    int s_off[CV_ROUNDS], s_dst[CV_ROUNDS], s_g8[CV_ROUNDS];
    bool s_ok[CV_ROUNDS];
    float s_m[CV_ROUNDS];
#pragma unroll
    for (int r = 0; r < CV_ROUNDS; ++r) {
        const int i = tid + r * CV_THREADS;
        const int g = i >= CV_NPX ? 1 : 0, p = i - g * CV_NPX;
        const int pr = p / CV_HW, pc = p - pr * CV_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        const bool live = i < CV_ITEMS;
        s_ok[r] = live & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        s_off[r] = s_ok[r] ? gy * a.W + gx : 0;
        s_g8[r] = g * 8;
        s_dst[r] = live ? g * CV_NPX + p : -1;
        s_m[r] = (pre == PRE_BN_MASK && s_ok[r]) ? a.pre_mask[(size_t)n * HW + s_off[r]] : 1.0f;
    }
    float st[CV_ROUNDS][8];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int r = 0; r < CV_ROUNDS; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j)                // channels past Cin re-read the last plane (zeroed below)
                st[r][j] = inb[(size_t)min(c * 16 + s_g8[r] + j, cmax) * HW + s_off[r]];
    };
    const bool has_pre = pre != PRE_NONE, nonzero_mask = pre == PRE_BN_NONZERO;
    auto store_chunk = [&](int buf, int c) {
#pragma unroll
        for (int r = 0; r < CV_ROUNDS; ++r) {
            if (s_dst[r] < 0) continue;
            const int cb = c * 16 + s_g8[r];
            // branch-free prologue (selects only): normalization.py:231, ReLU, partialconv2d.py:69
            float sc[8], sh[8];
            *reinterpret_cast<float4 *>(&sc[0]) = *reinterpret_cast<const float4 *>(&pss[0][cb]);
            *reinterpret_cast<float4 *>(&sc[4]) = *reinterpret_cast<const float4 *>(&pss[0][cb + 4]);
            *reinterpret_cast<float4 *>(&sh[0]) = *reinterpret_cast<const float4 *>(&pss[1][cb]);
            *reinterpret_cast<float4 *>(&sh[4]) = *reinterpret_cast<const float4 *>(&pss[1][cb + 4]);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = st[r][j];
                const float mk = nonzero_mask ? (x != 0.0f ? 1.0f : 0.0f) : s_m[r];     // s_m = 1 without a mask
                const float y = fmaxf(x * sc[j] - sh[j], 0.0f) * mk;
                v[j] = (s_ok[r] & (cb + j <= cmax)) ? (has_pre ? y : x) : 0.0f;       // zero padding (border, channels)
            }
            h8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = v[j] * CV_XSCALE;
                const _Float16 h = (_Float16)x;
                hi[j] = h;
                lo[j] = (_Float16)(x - (float)h);
            }
            (&xs[buf][0][0][0])[s_dst[r]] = hi;
            (&xs[buf][1][0][0])[s_dst[r]] = lo;
        }
    };

    f16v acc[CPW][PT];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.0f;

    load_chunk(0);
    __syncthreads();                                   // pss
    store_chunk(0, 0);
    __syncthreads();

    const int bcol = lane & 31, bgrp = lane >> 5;
    // A fragments (weights) come straight from global memory / L2, one tap ahead of their use.
    // Fragment (cot, chunk, tap, half) = 64 consecutive 16-byte vectors, lane l takes vector l.
    const size_t wtile = (size_t)nchunk * 9 * 2 * 64;            // vectors per 32-channel tile
    const h8 *wbase = a.w + (size_t)cot0 * wtile + lane;
    h8 a_cur[CPW][2], a_nxt[CPW][2];
    auto load_a = [&](h8 (&dst)[CPW][2], int g /* chunk * 9 + tap */) {
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) {
            const h8 *q = wbase + ct * wtile + (size_t)g * 128;
            dst[ct][0] = q[0];
            dst[ct][1] = q[64];
        }
    };
    load_a(a_cur, 0);
    const int glast = nchunk * 9 - 1;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk(c + 1);
        const h8 *xh = &xs[buf][0][bgrp][0], *xl = &xs[buf][1][bgrp][0];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            // keeps the B fragments of different taps from being kept live together (the 36 distinct
            // ones of a chunk would take 144 registers and spill); LDS has the bandwidth to re-read them
            asm volatile("" ::: "memory");
            load_a(a_nxt, min(c * 9 + tap + 1, glast));           // unconditional: counted vmcnt waits
            h8 bh[PT], bl[PT];
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = (wp * PT + pt + kh) * CV_HW + kw + bcol;
                bh[pt] = xh[p];
                bl[pt] = xl[p];
            }
            // the three partial products, each over all CPW x PT accumulator tiles: consecutive
            // MFMAs never touch the same accumulator (small terms first)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[ct][1], bh[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[ct][0], bl[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[ct][0], bh[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CPW; ++ct) { a_cur[ct][0] = a_nxt[ct][0]; a_cur[ct][1] = a_nxt[ct][1]; }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1, c + 1);
        __syncthreads();
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel).
    // Work-items outside the image / channels past Cout are clamped for the loads and skipped for
    // the stores; all loads of a tile are issued before its first store.
    // Straight-line code: absent operands are read through a valid stand-in address and selected
    // away, so the optional stages cost selects instead of branches.
    const bool partial = a.partial != 0, has_bias = a.bias != nullptr, has_res = a.residual != nullptr,
               has_next = a.next_scale != nullptr;
    const float *dummy = reinterpret_cast<const float *>(a.w);        // >= 144 * Cout floats
    const float *bp = has_bias ? a.bias : dummy;
    const float *nsp = has_next ? a.next_scale : dummy, *nhp = has_next ? a.next_shift : dummy;
    const float *mbp = partial ? a.mask_box : a.in;
    float *outp = a.out;
    const float *resid = has_res ? a.residual : outp;
    const int ox = x0 + bcol;
    const bool xin_img = ox < a.W;
    const int cout1 = a.Cout - 1;
    const float mscale = a.mask_scale, winsize = a.winsize, unscale = a.unscale;
#pragma unroll
    for (int ct = 0; ct < CPW; ++ct) {
        float eb[16], esc[16], esh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = min((cot0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp, cout1);
            const float b = bp[co], s2 = nsp[co], h2 = nhp[co];
            eb[r] = has_bias ? b : 0.0f;
            esc[r] = s2;
            esh[r] = h2;
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int oy = y0 + wp * PT + pt;
            const bool ok = xin_img & (oy < a.H);
            const size_t pix = ok ? (size_t)oy * a.W + ox : 0;
            const float u = mbp[(size_t)n * HW + pix] * mscale;            // partialconv2d.py:61-67
            const float um = fminf(fmaxf(u, 0.0f), 1.0f);
            const float ratio = (1.0f / (u + 1e-8f)) * winsize * um;       // torch: scalar / tensor = reciprocal * scalar
            if (partial && ok && a.um_out && ct == 0 && blockIdx.y == 0 && wc == 0 && bgrp == 0)
                a.um_out[(size_t)n * HW + pix] = um;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = min((cot0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp, cout1);
                rv[r] = resid[((size_t)n * a.Cout + co) * HW + pix];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (cot0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp;
                const float raw = acc[ct][pt][r] * unscale;
                float o = (raw * ratio + eb[r]) * um;                               // :72-74
                o = has_res ? o + rv[r] : o;                                        // blocks.py:248
                const float nx = fmaxf(o * esc[r] - esh[r], 0.0f) * um;             // blocks.py:233-236
                o = has_next ? nx : o;
                o = partial ? o : raw + eb[r];                                      // plain convolution (+ bias)
                if (ok && co <= cout1) outp[((size_t)n * a.Cout + co) * HW + pix] = o;
            }
        }
    }
}

// w [Cout,Cin,3,3] fp32 -> split f16 weights in fragment order over the PADDED channel counts
// (zero weights for the padding), scaled by wscale (a power of two)
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ ws,
                                                                 int Cout, int Cin, int CoutP, int CinP, float wscale) {
    const int total = CoutP * CinP * 9;
    const int nchunk = CinP >> 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % 9, ci = (i / 9) % CinP, co = i / (9 * CinP);
        const float x = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] * wscale : 0.0f;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const size_t frag = (((size_t)(co >> 5) * nchunk + (ci >> 4)) * 9 + tap) * 2;
        const int within = ((ci & 15) >> 3) * 256 + (co & 31) * 8 + (ci & 7);     // [ci group][co][8 ci]
        ws[frag * 512 + within] = h;
        ws[(frag + 1) * 512 + within] = l;
    }
}

}  // namespace slr

using namespace slr;

SLR_EXPORT size_t slr_conv3x3_weight_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)conv_cout_pad(Cout) * conv_cin_pad(Cin) * 9 * 2 * sizeof(_Float16);
}

SLR_EXPORT int slr_conv3x3_split_weights(const float *w, void *wsplit, int Cout, int Cin, float wscale, void *stream) {
    SLR_CHECK_ARG(w && wsplit, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cin > 0 && (long long)conv_cout_pad(Cout) * conv_cin_pad(Cin) * 9 < (1LL << 30), "sizes");
    SLR_CHECK_ARG(wscale > 0.0f, "wscale");
    const int CoutP = conv_cout_pad(Cout), CinP = conv_cin_pad(Cin);
    const int total = CoutP * CinP * 9;
    hipLaunchKernelGGL(conv_split_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16 *)wsplit, Cout, Cin, CoutP, CinP, wscale);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int conv_launch(ConvArgs &a, float wscale, hipStream_t st) {
    a.tiles_x = (a.W + CV_W - 1) / CV_W;
    a.nchunk = conv_cin_pad(a.Cin) / 16;
    a.unscale = 1.0f / (CV_XSCALE * wscale);
    const int tiles = a.tiles_x * ((a.H + CV_H - 1) / CV_H);
    const int ct = conv_cout_tile(a.Cout);
    const dim3 grid(tiles, conv_cout_pad(a.Cout) / ct, a.N);
    if (ct == 128) hipLaunchKernelGGL((conv3x3_split_kernel<2, 2>), grid, dim3(CV_THREADS), 0, st, a);
    else if (ct == 64) hipLaunchKernelGGL((conv3x3_split_kernel<2, 1>), grid, dim3(CV_THREADS), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_split_kernel<1, 1>), grid, dim3(CV_THREADS), 0, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}

static int conv_check_dims(int N, int Cin, int Cout, int H, int W) {
    SLR_CHECK_ARG(N > 0 && N < 65536 && Cin > 0 && Cout > 0 && Cout < (1 << 20) && H > 0 && W > 0 &&
                  (long long)Cin * H * W < (1LL << 31) && (long long)N * Cout * H * W < (1LL << 40), "sizes");
    return 0;
}

SLR_EXPORT int slr_conv3x3_forward(const float *in, const void *wsplit, const float *bias, float *out, int N, int Cin,
                                   int Cout, int H, int W, float wscale, const float *pre_scale,
                                   const float *pre_shift, void *stream) {
    SLR_CHECK_ARG(in && wsplit && out, "null pointer");
    SLR_CHECK_ARG(!pre_scale == !pre_shift, "pre_scale / pre_shift go together");
    SLR_CHECK_ARG(!pre_scale || Cin <= CV_MAXCIN, "prologue supports Cin <= 1024");
    if (int e = conv_check_dims(N, Cin, Cout, H, W)) return e;
    ConvArgs a = {};
    a.in = in; a.w = (const h8 *)wsplit; a.bias = bias; a.out = out;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.pre = pre_scale ? PRE_BN : PRE_NONE;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift;
    return conv_launch(a, wscale, (hipStream_t)stream);
}

SLR_EXPORT int slr_pconv3x3_forward(const float *x, const float *pre_scale, const float *pre_shift,
                                    const float *pre_mask, int pre_mask_mode, const void *wsplit, float wscale,
                                    const float *bias, const float *mask_box, float mask_scale,
                                    const float *residual, const float *next_scale, const float *next_shift,
                                    float *out, float *um_out, int N, int Cin, int Cout, int H, int W, void *stream) {
    SLR_CHECK_ARG(x && wsplit && bias && mask_box && out, "null pointer");
    SLR_CHECK_ARG(!pre_scale == !pre_shift, "pre_scale / pre_shift go together");
    SLR_CHECK_ARG(!pre_scale || Cin <= CV_MAXCIN, "prologue supports Cin <= 1024");
    SLR_CHECK_ARG(pre_mask_mode == -1 || pre_mask_mode == 0 || (pre_mask_mode == 1 && pre_mask), "pre_mask_mode");
    SLR_CHECK_ARG(pre_scale || pre_mask_mode == -1, "a prologue mask needs pre_scale / pre_shift");
    SLR_CHECK_ARG(!next_scale == !next_shift, "next_scale / next_shift go together");
    SLR_CHECK_ARG(!(residual && next_scale), "residual and next-BN fusion are exclusive");
    if (int e = conv_check_dims(N, Cin, Cout, H, W)) return e;
    ConvArgs a = {};
    a.in = x; a.w = (const h8 *)wsplit; a.bias = bias; a.out = out;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.pre = !pre_scale ? PRE_NONE : pre_mask_mode == 1 ? PRE_BN_MASK : pre_mask_mode == 0 ? PRE_BN_NONZERO : PRE_BN;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_mask = pre_mask;
    a.partial = 1;
    a.mask_box = mask_box; a.mask_scale = mask_scale; a.winsize = (float)Cin * 9.0f;
    a.residual = residual; a.next_scale = next_scale; a.next_shift = next_shift; a.um_out = um_out;
    return conv_launch(a, wscale, (hipStream_t)stream);
}
