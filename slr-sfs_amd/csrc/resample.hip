// resample.hip -- the memory-bound stages between the decoder's convolutions:
// 3x3 / stride 2 average pooling ("Down", models/layers/blocks.py:196-199 nn.AvgPool2d(3, 2, 1)),
// x2 bilinear up-sampling ("Up", blocks.py:200-203 nn.Upsample(scale_factor=2, mode='bilinear')),
// and the 1x1 skip convolution onto <= 4 output channels (blocks.py:192-193 with the 3-channel
// end of the decoder, configs.py:117-137).  One read of the input, one write of the output each;
// HBM-bound, no LDS needed (the 2x overlap between neighbouring work-items is served by L1/L2).
#include "slr_common.hpp"

namespace slr {

// out[n,c,oy,ox] = (1/9) * sum_{3x3} in[n,c,2oy-1+dy,2ox-1+dx], zero padded, padding counted in the
// divisor (count_include_pad=True, the nn.AvgPool2d default).  One work-item = two neighbouring output
// pixels = input columns 2*ox0-1 .. 2*ox0+3 of three rows; with VEC the four columns 2*ox0 .. 2*ox0+3
// come as one 16-byte load (W % 4 == 0), the left neighbour as a scalar.  Work-items are a flat index
// over the plane (no idle tail per row).
template <bool VEC>
__global__ __launch_bounds__(256) void avgpool3x3s2_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                           int H, int W, int OH, int OW) {
    const int OW2 = (OW + 1) / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW2) return;
    const int oy = idx / OW2, ox0 = (idx - oy * OW2) * 2;
    const size_t plane = blockIdx.y;
    const float *ip = in + plane * (size_t)H * W;
    float *op = out + (plane * OH + oy) * (size_t)OW;
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * oy - 1 + dy;
        const bool yin = (y >= 0) & (y < H);
        const float *row = ip + (size_t)(yin ? y : 0) * W;
        float v[5];
        const int xl = 2 * ox0 - 1;
        const float tl = row[xl >= 0 ? xl : 0];
        v[0] = (yin & (xl >= 0)) ? tl : 0.0f;
        if (VEC) {                                     // 2*ox0 is a multiple of 4 and 2*ox0 + 3 < W
            const float4 q = *reinterpret_cast<const float4 *>(row + 2 * ox0);
            v[1] = yin ? q.x : 0.0f; v[2] = yin ? q.y : 0.0f; v[3] = yin ? q.z : 0.0f; v[4] = yin ? q.w : 0.0f;
        } else {
#pragma unroll
            for (int k = 1; k < 5; ++k) {
                const int x = xl + k;
                const bool ok = yin & (x < W);
                const float t = row[ok ? x : 0];
                v[k] = ok ? t : 0.0f;
            }
        }
        s0 += (v[0] + v[1]) + v[2];
        s1 += (v[2] + v[3]) + v[4];
    }
    if (VEC) {
        *reinterpret_cast<float2 *>(op + ox0) = make_float2(s0 * (1.0f / 9.0f), s1 * (1.0f / 9.0f));
    } else {
        op[ox0] = s0 * (1.0f / 9.0f);
        if (ox0 + 1 < OW) op[ox0 + 1] = s1 * (1.0f / 9.0f);
    }
}

// x2 bilinear, align_corners=False: src = max((dst + 0.5) * 0.5 - 0.5, 0), i0 = floor(src),
// i1 = min(i0 + 1, in - 1), l1 = src - i0 (the formula of torch's upsample_bilinear2d).
// One work-item = input pixels (y, x), (y, x+1) -> the 2 x 4 output block below them, from a 3 x 4
// input neighbourhood (border rows / columns clamped: a clamped neighbour only ever meets weight 0).
__device__ __forceinline__ void up_axis(int o, int in_size, float &l0, float &l1) {
    const float s = fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int i0 = (int)s;
    (void)in_size;
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}
__global__ __launch_bounds__(256) void upsample2x_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                         int H, int W) {
    const int W2 = (W + 1) / 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W2) return;
    const int y = idx / W2, x = (idx - y * W2) * 2;
    const size_t plane = blockIdx.y;
    const float *ip = in + plane * (size_t)H * W;
    const int OW = 2 * W;
    float *op = out + (plane * 2 * H + 2 * y) * (size_t)OW + 2 * x;
    const int ys[3] = {max(y - 1, 0), y, min(y + 1, H - 1)};
    const int xs[4] = {max(x - 1, 0), x, min(x + 1, W - 1), min(x + 2, W - 1)};
    float v[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[r][k] = ip[(size_t)ys[r] * W + xs[k]];
    const bool two = x + 1 < W;                        // second input column exists (odd W: last work-item has one)
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
        float ly0, ly1;
        up_axis(2 * y + ry, H, ly0, ly1);
        const int ra = ry, rb = ry + 1;                // source rows (y-1, y) for the even output row, (y, y+1) for the odd one
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lx0, lx1;
            up_axis(2 * x + k, W, lx0, lx1);
            const int ca = (k + 1) / 2, cb = ca + 1;   // source columns: k=0: (x-1,x) 1: (x,x+1) 2: (x,x+1) 3: (x+1,x+2)
            o[k] = ly0 * (lx0 * v[ra][ca] + lx1 * v[ra][cb]) + ly1 * (lx0 * v[rb][ca] + lx1 * v[rb][cb]);
        }
        float *orow = op + (size_t)ry * OW;
        if (two && (OW % 4 == 0)) {
            *reinterpret_cast<float4 *>(orow) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            orow[0] = o[0]; orow[1] = o[1];
            if (two) { orow[2] = o[2]; orow[3] = o[3]; }
        }
    }
}

// 1x1 convolution onto COUT <= 4 channels: out[n,co,p] = b[co] + sum_ci w[co,ci] * in[n,ci,p].
// Four pixels per work-item (16-byte loads), weights through the scalar cache.
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const float4 *__restrict__ in, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float4 *__restrict__ out,
                                                            int Cin, int HW4) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW4) return;
    const float4 *ip = in + (size_t)n * Cin * HW4 + i;
    float4 acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { const float b = bias ? bias[co] : 0.0f; acc[co] = make_float4(b, b, b, b); }
#pragma unroll 8
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 v = ip[(size_t)ci * HW4];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float k = w[co * Cin + ci];
            acc[co].x = __builtin_fmaf(k, v.x, acc[co].x);
            acc[co].y = __builtin_fmaf(k, v.y, acc[co].y);
            acc[co].z = __builtin_fmaf(k, v.z, acc[co].z);
            acc[co].w = __builtin_fmaf(k, v.w, acc[co].w);
        }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) out[((size_t)n * COUT + co) * HW4 + i] = acc[co];
}

// ---- channel-blocked variants ([N, C/8, H, W, 8], see SLR_CONV_*_B8 in the header): a "pixel" is 8 floats = 32 bytes,
// one work-item per output pixel and 8-channel group, two 16-byte accesses per pixel.  Same per-channel arithmetic
// (and summation order) as the NCHW kernels above.
struct f8 { float4 a, b; };
__device__ __forceinline__ f8 ld8(const float *p) { const float4 *q = reinterpret_cast<const float4 *>(p); return {q[0], q[1]}; }
__device__ __forceinline__ void st8(float *p, const f8 &v) { float4 *q = reinterpret_cast<float4 *>(p); q[0] = v.a; q[1] = v.b; }

__global__ __launch_bounds__(256) void avgpool3x3s2_b8_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                              int H, int W, int OH, int OW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    const int oy = idx / OW, ox = idx - oy * OW;
    const size_t plane = blockIdx.y;
    const float *ip = in + plane * (size_t)H * W * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * oy - 1 + dy;
        const bool yin = (y >= 0) & (y < H);
        float v[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int x = 2 * ox - 1 + k;
            const bool ok = yin & (x >= 0) & (x < W);
            const f8 t = ld8(ip + ((size_t)(yin ? y : 0) * W + (ok ? x : 0)) * 8);
            const float tv[8] = {t.a.x, t.a.y, t.a.z, t.a.w, t.b.x, t.b.y, t.b.z, t.b.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) v[k][c] = ok ? tv[c] : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] += (v[0][c] + v[1][c]) + v[2][c];
    }
    f8 o;
    o.a = make_float4(s[0] * (1.0f / 9.0f), s[1] * (1.0f / 9.0f), s[2] * (1.0f / 9.0f), s[3] * (1.0f / 9.0f));
    o.b = make_float4(s[4] * (1.0f / 9.0f), s[5] * (1.0f / 9.0f), s[6] * (1.0f / 9.0f), s[7] * (1.0f / 9.0f));
    st8(out + (plane * OH * OW + idx) * 8, o);
}

__global__ __launch_bounds__(256) void upsample2x_b8_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                            int H, int W) {
    const int OW = 2 * W, OH = 2 * H;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    const int oy = idx / OW, ox = idx - oy * OW;
    const size_t plane = blockIdx.y;
    float ly0, ly1, lx0, lx1;
    up_axis(oy, H, ly0, ly1);
    up_axis(ox, W, lx0, lx1);
    const float sy = fmaxf((oy + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf((ox + 0.5f) * 0.5f - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float *ip = in + plane * (size_t)H * W * 8;
    const f8 p00 = ld8(ip + ((size_t)y0 * W + x0) * 8), p01 = ld8(ip + ((size_t)y0 * W + x1) * 8);
    const f8 p10 = ld8(ip + ((size_t)y1 * W + x0) * 8), p11 = ld8(ip + ((size_t)y1 * W + x1) * 8);
    const float a00[8] = {p00.a.x, p00.a.y, p00.a.z, p00.a.w, p00.b.x, p00.b.y, p00.b.z, p00.b.w};
    const float a01[8] = {p01.a.x, p01.a.y, p01.a.z, p01.a.w, p01.b.x, p01.b.y, p01.b.z, p01.b.w};
    const float a10[8] = {p10.a.x, p10.a.y, p10.a.z, p10.a.w, p10.b.x, p10.b.y, p10.b.z, p10.b.w};
    const float a11[8] = {p11.a.x, p11.a.y, p11.a.z, p11.a.w, p11.b.x, p11.b.y, p11.b.z, p11.b.w};
    float r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = ly0 * (lx0 * a00[c] + lx1 * a01[c]) + ly1 * (lx0 * a10[c] + lx1 * a11[c]);
    f8 o;
    o.a = make_float4(r[0], r[1], r[2], r[3]);
    o.b = make_float4(r[4], r[5], r[6], r[7]);
    st8(out + (plane * (size_t)OH * OW + idx) * 8, o);
}

// 1x1 convolution onto COUT <= 4 channels from a channel-blocked input: one pixel per work-item, channels in
// ascending order (the NCHW kernel's order per pixel: bit-identical results).
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_b8_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               int Cin, int HW) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *ip = in + ((size_t)n * (Cin >> 3) * HW + i) * 8;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.0f;
    for (int g = 0; g < (Cin >> 3); ++g) {
        const f8 t = ld8(ip + (size_t)g * HW * 8);
        const float tv[8] = {t.a.x, t.a.y, t.a.z, t.a.w, t.b.x, t.b.y, t.b.z, t.b.w};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = __builtin_fmaf(w[co * Cin + g * 8 + c], tv[c], acc[co]);
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) out[((size_t)n * COUT + co) * HW + i] = acc[co];
}

}  // namespace slr

using namespace slr;

SLR_EXPORT int slr_avgpool3x3s2(const float *in, float *out, int N, int C, int H, int W, int b8, void *stream) {
    SLR_CHECK_ARG(in && out, "null pointer");
    SLR_CHECK_ARG(!b8 || (C % 8 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 15)), "channel-blocked layout needs C % 8 == 0 and 16-byte aligned tensors");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (long long)N * C < 65536 && (long long)H * W < (1LL << 30), "sizes");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    if (b8) {
        hipLaunchKernelGGL(avgpool3x3s2_b8_kernel, dim3((OH * OW + 255) / 256, N * C / 8), dim3(256), 0, (hipStream_t)stream, in, out, H, W, OH, OW);
        SLR_CHECK_LAUNCH();
        return 0;
    }
    const int items = OH * ((OW + 1) / 2);
    const dim3 grid((items + 255) / 256, N * C);
    const bool vec = (W % 4 == 0) && !(((uintptr_t)in | (uintptr_t)out) & 15);       // then OW is even, too
    if (vec) hipLaunchKernelGGL(avgpool3x3s2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W, OH, OW);
    else hipLaunchKernelGGL(avgpool3x3s2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, out, H, W, OH, OW);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_upsample_bilinear2x(const float *in, float *out, int N, int C, int H, int W, int b8, void *stream) {
    SLR_CHECK_ARG(in && out, "null pointer");
    SLR_CHECK_ARG(!b8 || (C % 8 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 15)), "channel-blocked layout needs C % 8 == 0 and 16-byte aligned tensors");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (long long)N * C < 65536 && (long long)H * W < (1LL << 28), "sizes");
    SLR_CHECK_ARG(!((uintptr_t)out & 15), "16-byte aligned output");
    if (b8) {
        hipLaunchKernelGGL(upsample2x_b8_kernel, dim3((4 * H * W + 255) / 256, N * C / 8), dim3(256), 0, (hipStream_t)stream, in, out, H, W);
        SLR_CHECK_LAUNCH();
        return 0;
    }
    const int items = H * ((W + 1) / 2);
    hipLaunchKernelGGL(upsample2x_kernel, dim3((items + 255) / 256, N * C), dim3(256), 0, (hipStream_t)stream, in, out, H, W);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_conv1x1_small(const float *in, const float *w, const float *bias, float *out, int N, int Cin,
                                 int Cout, int H, int W, int in_b8, void *stream) {
    SLR_CHECK_ARG(in && w && out, "null pointer");
    SLR_CHECK_ARG(Cout >= 1 && Cout <= 4, "1 <= Cout <= 4");
    SLR_CHECK_ARG(N > 0 && N < 65536 && Cin > 0 && H > 0 && W > 0, "sizes");
    if (in_b8) {
        SLR_CHECK_ARG(Cin % 8 == 0 && !((uintptr_t)in & 15), "channel-blocked input needs Cin % 8 == 0 and a 16-byte aligned tensor");
        const dim3 g8((H * W + 255) / 256, N);
        hipStream_t s8 = (hipStream_t)stream;
#define LAUNCH8(CO) hipLaunchKernelGGL(conv1x1_small_b8_kernel<CO>, g8, dim3(256), 0, s8, in, w, bias, out, Cin, H * W)
        if (Cout == 1) LAUNCH8(1); else if (Cout == 2) LAUNCH8(2); else if (Cout == 3) LAUNCH8(3); else LAUNCH8(4);
#undef LAUNCH8
        SLR_CHECK_LAUNCH();
        return 0;
    }
    SLR_CHECK_ARG(((size_t)H * W) % 4 == 0 && !(((uintptr_t)in | (uintptr_t)out) & 15), "H*W % 4 == 0 and 16-byte aligned tensors");
    const int HW4 = H * W / 4;
    const dim3 grid((HW4 + 255) / 256, N);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(CO) hipLaunchKernelGGL(conv1x1_small_kernel<CO>, grid, dim3(256), 0, st, (const float4 *)in, w, bias, (float4 *)out, Cin, HW4)
    if (Cout == 1) LAUNCH(1); else if (Cout == 2) LAUNCH(2); else if (Cout == 3) LAUNCH(3); else LAUNCH(4);
#undef LAUNCH
    SLR_CHECK_LAUNCH();
    return 0;
}
