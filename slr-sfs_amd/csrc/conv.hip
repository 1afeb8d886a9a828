// conv.hip -- 3x3 / stride 1 / pad 1 convolution of the partial-convolution decoder on the matrix
// cores (gfx950), fp32 in / fp32 out, as an implicit GEMM on v_mfma_f32_32x32x16_f16 with SPLIT
// operands (models/layers/partialconv2d.py:69 `raw_out = conv(input * mask)`,
// models/layers/blocks.py:233-239; 95 % of a frame's time once the splat is fused).
//
// gfx950 has no TF32-like mode: fp32-input MFMA runs at the fp32 vector rate (157 TFLOP/s), 1/16
// of the f16 rate.  Every fp32 operand is therefore split into two halves, x = hi + lo with
// hi = f16(x), lo = f16(x - hi) (22 significant bits), and the product is formed from three f16
// MFMAs with fp32 accumulation:  x*w ~= hi*hi + hi*lo + lo*hi  (the dropped lo*lo term is
// 2^-22 relative).  Both operands are pre-scaled by powers of two so the lo halves stay normal
// f16 numbers; the inverse scale is applied (exactly) to the accumulators at the end.
// Measured error of the whole decoder vs an fp64 convolution: same class as MIOpen's fp32 Winograd.
//
// Data flow of one workgroup (256 work-items = 4 waves, output block = 8 rows x 32 columns x
// (64 | 128) output channels of one sample):
//   * per chunk of 16 input channels the (8+2) x (32+2) input halo block is loaded from the NCHW
//     tensor (coalesced along x), split, and stored in LDS as [half][8-channel group][pixel] 16-byte
//     vectors -- exactly the B fragment of the MFMA (lane l: pixel l&31, channels 8*(l>>5)..+7),
//     so fragment reads are conflict-free ds_read_b128 and the 9 taps are just shifted pixel offsets;
//     double-buffered: the loads of chunk c+1 are in flight under the MFMAs of chunk c, one
//     barrier per chunk;
//   * the pre-split weights are stored on the host side in fragment order
//     [co/32][ci/16][tap][half][ci group 2][co 32][ci 8]; a wave reads its A fragments straight from
//     global memory (1 KiB contiguous per fragment, L2-resident: 0.6 MB for 128x128 channels);
//   * wave (wc, wp) owns output channels [wc*32*CPW, +32*CPW) and rows [4*wp, 4*wp+4): CPW x 4
//     accumulator tiles of 32x32 (128 registers at CPW = 2), 3 MFMAs per tile, tap and chunk.
// The D layout (row = channel, column = pixel) makes every accumulator register a pair of 128-byte
// row segments of the NCHW output.
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

struct ConvArgs {
    const float *in;       // [N,Cin,H,W]
    const h8 *w;           // split weights in fragment order (see above), scaled by wscale
    const float *bias;     // [Cout] or nullptr
    float *out;            // [N,Cout,H,W]
    int N, Cin, Cout, H, W, tiles_x;
    float unscale;         // 1 / (CV_XSCALE * wscale)
};

__device__ __forceinline__ void split8(const float (&v)[8], h8 &hi, h8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * CV_XSCALE;
        const _Float16 h = (_Float16)x;
        hi[j] = h;
        lo[j] = (_Float16)(x - (float)h);
    }
}

// CPW: 32-channel output tiles per wave (2: workgroup covers 128 output channels, 1: 64).
template <int CPW>
__global__ __launch_bounds__(CV_THREADS, 2) void conv3x3_split_kernel(ConvArgs a) {
    __shared__ h8 xs[2][2][2][CV_NPX];             // [buffer][hi|lo][8-channel group][halo pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const int x0 = tx * CV_W, y0 = ty * CV_H;
    const int n = blockIdx.z;
    const int cot0 = (blockIdx.y * 2 + wc) * CPW;  // first 32-channel tile of this wave
    const int HW = a.H * a.W;
    const int nchunk = a.Cin >> 4;
    const float *inb = a.in + (size_t)n * a.Cin * HW;

    // staging items of this work-item: (halo pixel, 8-channel group), constant over the chunks
    int s_off[CV_ROUNDS], s_dst[CV_ROUNDS];
    bool s_ok[CV_ROUNDS];
#pragma unroll
    for (int r = 0; r < CV_ROUNDS; ++r) {
        const int i = tid + r * CV_THREADS;
        const int g = i >= CV_NPX ? 1 : 0, p = i - g * CV_NPX;
        const int pr = p / CV_HW, pc = p - pr * CV_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        const bool live = i < CV_ITEMS;
        s_ok[r] = live & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        s_off[r] = s_ok[r] ? g * 8 * HW + gy * a.W + gx : 0;
        s_dst[r] = live ? g * CV_NPX + p : -1;
    }
    float st[CV_ROUNDS][8];
    auto load_chunk = [&](int c) {
        const float *pl = inb + (size_t)c * 16 * HW;
#pragma unroll
        for (int r = 0; r < CV_ROUNDS; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) st[r][j] = pl[s_off[r] + j * HW];
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int r = 0; r < CV_ROUNDS; ++r) {
            if (s_dst[r] < 0) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s_ok[r] ? st[r][j] : 0.0f;     // zero padding
            h8 hi, lo;
            split8(v, hi, lo);
            (&xs[buf][0][0][0])[s_dst[r]] = hi;
            (&xs[buf][1][0][0])[s_dst[r]] = lo;
        }
    };

    f16v acc[CPW][4];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][p][r] = 0.0f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int bcol = lane & 31, bgrp = lane >> 5;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) load_chunk(c + 1);
        const h8 *xh = &xs[buf][0][bgrp][0], *xl = &xs[buf][1][bgrp][0];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            h8 ah[CPW], al[CPW];
#pragma unroll
            for (int ct = 0; ct < CPW; ++ct) {
                const h8 *wp_ = a.w + ((((size_t)(cot0 + ct) * nchunk + c) * 9 + tap) * 2) * 64 + lane;
                ah[ct] = wp_[0];
                al[ct] = wp_[64];
            }
            h8 bh[4], bl[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int p = (wp * 4 + pt + kh) * CV_HW + kw + bcol;
                bh[pt] = xh[p];
                bl[pt] = xl[p];
            }
            // the three partial products, each over all CPW x 4 accumulator tiles: consecutive
            // MFMAs never touch the same accumulator (small terms first)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ct], bh[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bl[pt], acc[ct][pt], 0, 0, 0);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < CPW; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bh[pt], acc[ct][pt], 0, 0, 0);
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
    const int ox = x0 + bcol;
    if (ox < a.W) {
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                const int oy = y0 + wp * 4 + pt;
                if (oy >= a.H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (cot0 + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * bgrp;
                    float v = acc[ct][pt][r] * a.unscale;
                    if (a.bias) v += a.bias[co];
                    a.out[((size_t)n * a.Cout + co) * HW + (size_t)oy * a.W + ox] = v;
                }
            }
    }
}

// w [Cout,Cin,3,3] fp32 -> split f16 weights in fragment order, scaled by wscale (a power of two)
__global__ __launch_bounds__(256) void conv_split_weights_kernel(const float *__restrict__ w, _Float16 *__restrict__ ws,
                                                                 int Cout, int Cin, float wscale) {
    const int total = Cout * Cin * 9;
    const int nchunk = Cin >> 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % 9, ci = (i / 9) % Cin, co = i / (9 * Cin);
        const float x = w[i] * wscale;
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

SLR_EXPORT size_t slr_conv3x3_weight_bytes(int Cout, int Cin) { return (size_t)Cout * Cin * 9 * 2 * sizeof(_Float16); }

SLR_EXPORT int slr_conv3x3_split_weights(const float *w, void *wsplit, int Cout, int Cin, float wscale, void *stream) {
    SLR_CHECK_ARG(w && wsplit, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cout % 64 == 0 && Cin > 0 && Cin % 16 == 0, "Cout % 64 == 0 and Cin % 16 == 0 required");
    SLR_CHECK_ARG(wscale > 0.0f, "wscale");
    const int total = Cout * Cin * 9;
    hipLaunchKernelGGL(conv_split_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (_Float16 *)wsplit, Cout, Cin, wscale);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_conv3x3_forward(const float *in, const void *wsplit, const float *bias, float *out, int N, int Cin,
                                   int Cout, int H, int W, float wscale, void *stream) {
    SLR_CHECK_ARG(in && wsplit && out, "null pointer");
    SLR_CHECK_ARG(Cout > 0 && Cout % 64 == 0 && Cin > 0 && Cin % 16 == 0, "Cout % 64 == 0 and Cin % 16 == 0 required");
    SLR_CHECK_ARG(N > 0 && N < 65536 && H > 0 && W > 0 && (long long)Cin * H * W < (1LL << 31) &&
                  (long long)N * Cout * H * W < (1LL << 40), "sizes");
    ConvArgs a;
    a.in = in; a.w = (const h8 *)wsplit; a.bias = bias; a.out = out;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    a.tiles_x = (W + CV_W - 1) / CV_W;
    a.unscale = 1.0f / (CV_XSCALE * wscale);
    const int tiles = a.tiles_x * ((H + CV_H - 1) / CV_H);
    hipStream_t st = (hipStream_t)stream;
    if (Cout % 128 == 0)
        hipLaunchKernelGGL(conv3x3_split_kernel<2>, dim3(tiles, Cout / 128, N), dim3(CV_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL(conv3x3_split_kernel<1>, dim3(tiles, Cout / 64, N), dim3(CV_THREADS), 0, st, a);
    SLR_CHECK_LAUNCH();
    return 0;
}
