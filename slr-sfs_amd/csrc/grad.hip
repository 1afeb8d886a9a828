// grad.hip -- backward of the summation splat and the inverse (gather) half of the
// maximum-warp-norm splat.  All pure gathers: one work-item per source PIXEL, the flow and the
// four corner weights are computed once and reused for every channel (the reference recomputes
// them per element / per thread, models/softsplat.py:204-326, 84-155).
#include "slr_common.hpp"

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
#ifndef SLR_GRAD_U
#define SLR_GRAD_U 4
#endif

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

SLR_EXPORT int slr_softsplat_backward(const float *in, const float *flow, const float *grad_out, float *grad_in,
                                      float *grad_flow, int N, int C, int H, int W, void *stream) {
    SLR_CHECK_ARG(flow && grad_out, "null pointer");
    SLR_CHECK_ARG(!grad_flow || in, "input required for grad_flow");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && (long long)N * H * W < (1LL << 29), "sizes");
    dim3 grid((H * W + 255) / 256, N);
    hipStream_t st = (hipStream_t)stream;
    if (grad_in && grad_flow) hipLaunchKernelGGL((grad_kernel<true, true>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
    else if (grad_in) hipLaunchKernelGGL((grad_kernel<true, false>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
    else if (grad_flow) hipLaunchKernelGGL((grad_kernel<false, true>), grid, dim3(256), 0, st, in, flow, grad_out, grad_in, grad_flow, C, H, W);
    SLR_CHECK_LAUNCH();
    return 0;
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
