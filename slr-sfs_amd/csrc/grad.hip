// grad.hip -- backward of the summation splat and the inverse (gather) half of the
// maximum-warp-norm splat.  All pure gathers: one work-item per source PIXEL, the flow and the
// four corner weights are computed once and reused for every channel (the reference recomputes
// them per element / per thread, models/softsplat.py:204-326, 84-155).
#include "slr_common.hpp"

namespace slr {

// gradInput[n,c,y,x] = sum_corners gradOutput[n,c,corner] * w   (softsplat.py:204-255)
// Sum order NW, NE, SW, SE and no FMA contraction (built with -ffp-contract=off): bit-exact
// with the reference kernel.
__global__ __launch_bounds__(256) void grad_input_kernel(const float *__restrict__ flow,
                                                         const float *__restrict__ gout,
                                                         float *__restrict__ gin, int C, int H, int W) {
    const int HW = H * W;
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *f = flow + (size_t)n * 2 * HW;
    const int y = i / W, x = i - y * W;
    const Corners c = make_corners(f[i], f[HW + i], x, y);
    const bool k0 = c.ok & in_image(c.x0, c.y0, H, W), k1 = c.ok & in_image(c.x0 + 1, c.y0, H, W);
    const bool k2 = c.ok & in_image(c.x0, c.y0 + 1, H, W), k3 = c.ok & in_image(c.x0 + 1, c.y0 + 1, H, W);
    // Branch-free channel loop: out-of-image corners read a valid address (this pixel) and their
    // PRODUCT is replaced by +0.0, so the sum has the reference's terms in the reference's order
    // (adding +0.0 changes nothing but the sign of a -0.0) and the loads of several channels overlap.
    const int o = c.y0 * W + c.x0;
    const int o0 = k0 ? o : i, o1 = k1 ? o + 1 : i, o2 = k2 ? o + W : i, o3 = k3 ? o + W + 1 : i;
    const float *gp = gout + (size_t)n * C * HW;
    float *op = gin + (size_t)n * C * HW;
#pragma unroll 4
    for (int ch = 0; ch < C; ++ch, gp += HW, op += HW) {
        const float a0 = gp[o0], a1 = gp[o1], a2 = gp[o2], a3 = gp[o3];
        float g = 0.0f;
        g += k0 ? a0 * c.w[0] : 0.0f;
        g += k1 ? a1 * c.w[1] : 0.0f;
        g += k2 ? a2 * c.w[2] : 0.0f;
        g += k3 ? a3 * c.w[3] : 0.0f;
        op[i] = g;
    }
}

// gradFlow[n,{x,y},y,x] = sum_c in[c] * sum_corners gradOutput[c,corner] * dw/d{x,y}
// (softsplat.py:257-326).  The reference runs one thread per flow COMPONENT with a C-loop
// each; here one work-item produces both components from one pass over the channels, with the
// same accumulation order per component ((in*gout)*dw; channels outer, corners NW..SE inner).
__global__ __launch_bounds__(256) void grad_flow_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ flow,
                                                        const float *__restrict__ gout,
                                                        float *__restrict__ gflow, int C, int H, int W) {
    const int HW = H * W;
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const float *f = flow + (size_t)n * 2 * HW;
    const int y = i / W, x = i - y * W;
    const float X = (float)x + f[i], Y = (float)y + f[HW + i];
    const bool ok = (fabsf(X) < 1073741824.0f) && (fabsf(Y) < 1073741824.0f);
    const int x0 = ok ? (int)floorf(X) : 0, y0 = ok ? (int)floorf(Y) : 0;
    const float ax = (float)(x0 + 1) - X, bx = X - (float)x0;      // d/dy weights use these
    const float ay = (float)(y0 + 1) - Y, by = Y - (float)y0;      // d/dx weights use these
    // softsplat.py:289-299
    const float dx[4] = {(-1.0f) * ay, (+1.0f) * ay, (-1.0f) * by, (+1.0f) * by};
    const float dy[4] = {ax * (-1.0f), bx * (-1.0f), ax * (+1.0f), bx * (+1.0f)};
    const bool k0 = ok & in_image(x0, y0, H, W), k1 = ok & in_image(x0 + 1, y0, H, W);
    const bool k2 = ok & in_image(x0, y0 + 1, H, W), k3 = ok & in_image(x0 + 1, y0 + 1, H, W);
    const int o = y0 * W + x0;
    const int o0 = k0 ? o : i, o1 = k1 ? o + 1 : i, o2 = k2 ? o + W : i, o3 = k3 ? o + W + 1 : i;   // see grad_input_kernel
    const float *ip = in + (size_t)n * C * HW;
    const float *gp = gout + (size_t)n * C * HW;
    float gx = 0.0f, gy = 0.0f;
#pragma unroll 4
    for (int ch = 0; ch < C; ++ch, ip += HW, gp += HW) {
        const float v = ip[i];
        const float t0 = v * gp[o0], t1 = v * gp[o1], t2 = v * gp[o2], t3 = v * gp[o3];
        gx += k0 ? t0 * dx[0] : 0.0f; gy += k0 ? t0 * dy[0] : 0.0f;
        gx += k1 ? t1 * dx[1] : 0.0f; gy += k1 ? t1 * dy[1] : 0.0f;
        gx += k2 ? t2 * dx[2] : 0.0f; gy += k2 ? t2 * dy[2] : 0.0f;
        gx += k3 ? t3 * dx[3] : 0.0f; gy += k3 ? t3 * dy[3] : 0.0f;
    }
    gflow[(size_t)n * 2 * HW + i] = gx;
    gflow[(size_t)n * 2 * HW + HW + i] = gy;
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
    if (grad_in) hipLaunchKernelGGL(grad_input_kernel, grid, dim3(256), 0, st, flow, grad_out, grad_in, C, H, W);
    if (grad_flow) hipLaunchKernelGGL(grad_flow_kernel, grid, dim3(256), 0, st, in, flow, grad_out, grad_flow, C, H, W);
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
