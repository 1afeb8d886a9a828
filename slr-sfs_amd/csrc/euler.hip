// euler.hip -- Euler integration of a static Eulerian motion field (gfx950).
//
// Replaces models/projection/euler_integration_manipulator.py:7-56: the reference runs a
// Python loop of ~15 tiny torch kernels per step (with a boolean-mask index_put that syncs
// the host), re-started from scratch for every frame.  Pixels are independent -- each step
// only GATHERS from the static field -- so one work-item integrates one pixel for all steps
// with its coordinate in registers.  The field (2 planes, 7.9 MB at 768x1280) stays in
// L2 / Infinity Cache; the kernel is latency-bound on the dependent gather chain, so it runs
// at full occupancy with 256-thread workgroups and lanes mapped to consecutive x (coalesced
// plane stores for the all-frames variant).
#include "slr_common.hpp"

namespace slr {

// One reference step (euler_integration_manipulator.py:37-46).  Bit-exact: fp32 adds only
// (sign is +-1, so sign*m is exact), rintf == torch.round (half to even).
__device__ __forceinline__ void euler_step(const float *__restrict__ mx, const float *__restrict__ my,
                                           int H, int W, float sign, float ox, float oy,
                                           float &px, float &py, bool &inv) {
    int ix = (int)rintf(px);
    int iy = (int)rintf(py);
    int g = iy * W + ix;
    float nx = px + sign * mx[g];
    float ny = py + sign * my[g];
    bool oob = (nx > (float)(W - 1)) | (nx < 0.0f) | (ny > (float)(H - 1)) | (ny < 0.0f) |
               !(nx == nx) | !(ny == ny);
    inv |= oob;
    px = inv ? ox : nx;
    py = inv ? oy : ny;
}

// grid.y = sample of a batch; steps (device, one count per sample) overrides nsteps when given: the batch form of
// EulerIntegration.forward (euler_integration_manipulator.py:58-71) is ONE launch with no step count read back by the host.
__global__ __launch_bounds__(256) void euler_kernel(const float *__restrict__ motion, int H, int W,
                                                    int nsteps, const long long *__restrict__ steps, float sign,
                                                    float *__restrict__ disp, float *__restrict__ visible) {
    const int HW = H * W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const size_t b = blockIdx.y;
    motion += b * 2 * HW; disp += b * 2 * HW;
    if (visible) visible += b * HW;
    if (steps) nsteps = (int)min(max(steps[b], 0ll), (long long)0x7fffffff);      // (range(1, n + 1): no step for n <= 0)
    const float *mx = motion, *my = motion + HW;
    const int y = i / W, x = i - y * W;
    const float ox = (float)x, oy = (float)y;
    float px = ox, py = oy;
    bool inv = false;
    for (int s = 0; s < nsteps; ++s) {
        euler_step(mx, my, H, W, sign, ox, oy, px, py, inv);
        if (inv) break;                       // sticky: later steps cannot change the result
    }
    const float big = (float)(H > W ? H : W) + 1.0f;
    disp[i] = inv ? big : px - ox;
    disp[HW + i] = inv ? big : py - oy;
    if (visible) visible[i] = inv ? 0.0f : 1.0f;
}

// All frames t = 0 .. nmax of one direction in one pass; a work-item carries SLR_EULER_PPT pixels (256 apart: every store of a wave is
// 256 contiguous bytes).  The displacement maps are written once and read by later kernels only: nontemporal stores (round 5: 135 ->
// 115-127 us per direction at 768x1280, N = 60 = 480 MB written: ~4 TB/s, store-bound).  More independent gather chains per lane do
// not help -- 2 pixels per work-item 125-134 us, 4: 144-151 (the chain's latency is hidden already; fewer waves store less evenly).
#define SLR_EULER_PPT 1
__global__ __launch_bounds__(256) void euler_all_kernel(const float *__restrict__ motion, int H, int W,
                                                        int nmax, float sign,
                                                        float *__restrict__ disp_all,
                                                        float *__restrict__ vis_all) {
    constexpr int P = SLR_EULER_PPT;
    const int HW = H * W;
    const int i0 = blockIdx.x * (256 * P) + threadIdx.x;
    const float *mx = motion, *my = motion + HW;
    const float big = (float)(H > W ? H : W) + 1.0f;
    float ox[P], oy[P], px[P], py[P];
    bool inv[P], on[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int i = i0 + 256 * k;
        on[k] = i < HW;
        const int ii = on[k] ? i : 0, y = ii / W, x = ii - y * W;
        ox[k] = px[k] = (float)x; oy[k] = py[k] = (float)y;
        inv[k] = false;
    }
    for (int t = 0; t <= nmax; ++t) {
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (t > 0 && !inv[k]) euler_step(mx, my, H, W, sign, ox[k], oy[k], px[k], py[k], inv[k]);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (!on[k]) continue;
            const size_t o = (size_t)t * 2 * HW + i0 + 256 * k;
            __builtin_nontemporal_store(inv[k] ? big : px[k] - ox[k], disp_all + o);
            __builtin_nontemporal_store(inv[k] ? big : py[k] - oy[k], disp_all + o + HW);
            if (vis_all) vis_all[(size_t)t * HW + i0 + 256 * k] = inv[k] ? 0.0f : 1.0f;
        }
    }
}

// Backward of euler_kernel w.r.t. the motion field -- what torch autograd computes through the reference's loop
// (euler_integration_manipulator.py:36-55): the gather at the rounded coordinate (:37-38) is the only differentiable
// use of `motion`, so each step of a pixel's path receives the pixel's output gradient; a pixel that ever leaves the
// image has its coordinate and its displacement overwritten with constants (:45-46, :55) -> no gradient.
// One work-item per pixel: pass 1 finds out whether the path stays valid, pass 2 re-walks it and scatters (global fp32
// atomics: many paths cross the same cell; training-only path, HBM-atomic-bound, order as unspecified as torch's own
// index_put_(accumulate=True) backward).
__global__ __launch_bounds__(256) void euler_backward_kernel(const float *__restrict__ motion, int H, int W, int nsteps,
                                                             const long long *__restrict__ steps,
                                                             float sign, const float *__restrict__ gdisp,
                                                             float *__restrict__ gmotion) {
    const int HW = H * W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const size_t b = blockIdx.y;                          // sample of a batch (see euler_kernel)
    motion += b * 2 * HW; gdisp += b * 2 * HW; gmotion += b * 2 * HW;
    if (steps) nsteps = (int)min(max(steps[b], 0ll), (long long)0x7fffffff);
    const float *mx = motion, *my = motion + HW;
    const int y = i / W, x = i - y * W;
    const float ox = (float)x, oy = (float)y;
    float px = ox, py = oy;
    bool inv = false;
    for (int s = 0; s < nsteps && !inv; ++s) euler_step(mx, my, H, W, sign, ox, oy, px, py, inv);
    if (inv) return;
    const float gx = sign * gdisp[i], gy = sign * gdisp[HW + i];
    px = ox; py = oy;
    for (int s = 0; s < nsteps; ++s) {
        const int g = (int)rintf(py) * W + (int)rintf(px);
        atomicAdd(&gmotion[g], gx);
        atomicAdd(&gmotion[HW + g], gy);
        euler_step(mx, my, H, W, sign, ox, oy, px, py, inv);
    }
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float *__restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

}  // namespace slr

SLR_EXPORT int slr_euler_integrate(const float *motion, int H, int W, int nsteps, float sign,
                                   float *disp, float *visible, void *stream) {
    SLR_CHECK_ARG(motion && disp, "null pointer");
    SLR_CHECK_ARG(H > 0 && W > 0 && nsteps >= 0, "sizes");
    SLR_CHECK_ARG((long long)H * W < (1LL << 30), "H*W too large");
    SLR_CHECK_ARG(sign == 1.0f || sign == -1.0f, "sign must be +1 or -1");
    int blocks = (H * W + 255) / 256;
    hipLaunchKernelGGL(slr::euler_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       motion, H, W, nsteps, (const long long *)nullptr, sign, disp, visible);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_euler_integrate_batch(const float *motion, const long long *steps, int B, int H, int W, float sign,
                                         float *disp, float *visible, void *stream) {
    SLR_CHECK_ARG(motion && steps && disp, "null pointer");
    SLR_CHECK_ARG(B > 0 && B < 65536 && H > 0 && W > 0, "sizes");
    SLR_CHECK_ARG((long long)H * W < (1LL << 30), "H*W too large");
    SLR_CHECK_ARG(sign == 1.0f || sign == -1.0f, "sign must be +1 or -1");
    hipLaunchKernelGGL(slr::euler_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       motion, H, W, 0, steps, sign, disp, visible);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_euler_integrate_all(const float *motion, int H, int W, int nmax, float sign,
                                       float *disp_all, float *vis_all, void *stream) {
    SLR_CHECK_ARG(motion && disp_all, "null pointer");
    SLR_CHECK_ARG(H > 0 && W > 0 && nmax >= 0, "sizes");
    SLR_CHECK_ARG((long long)H * W < (1LL << 30), "H*W too large");
    SLR_CHECK_ARG(sign == 1.0f || sign == -1.0f, "sign must be +1 or -1");
    int blocks = (H * W + 256 * SLR_EULER_PPT - 1) / (256 * SLR_EULER_PPT);
    hipLaunchKernelGGL(slr::euler_all_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       motion, H, W, nmax, sign, disp_all, vis_all);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_euler_backward(const float *motion, int H, int W, int nsteps, float sign, const float *grad_disp,
                                  float *grad_motion, void *stream) {
    SLR_CHECK_ARG(motion && grad_disp && grad_motion, "null pointer");
    SLR_CHECK_ARG(H > 0 && W > 0 && nsteps >= 0, "sizes");
    SLR_CHECK_ARG((long long)H * W < (1LL << 30), "H*W too large");
    SLR_CHECK_ARG(sign == 1.0f || sign == -1.0f, "sign must be +1 or -1");
    const int blocks = (H * W + 255) / 256;
    hipLaunchKernelGGL(slr::zero_f32_kernel, dim3(2 * blocks), dim3(256), 0, (hipStream_t)stream, grad_motion,
                       (size_t)2 * H * W);
    hipLaunchKernelGGL(slr::euler_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, motion, H, W, nsteps,
                       (const long long *)nullptr, sign, grad_disp, grad_motion);
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_euler_backward_batch(const float *motion, const long long *steps, int B, int H, int W, float sign,
                                        const float *grad_disp, float *grad_motion, void *stream) {
    SLR_CHECK_ARG(motion && steps && grad_disp && grad_motion, "null pointer");
    SLR_CHECK_ARG(B > 0 && B < 65536 && H > 0 && W > 0, "sizes");
    SLR_CHECK_ARG((long long)H * W < (1LL << 30) && (long long)B * H * W < (1LL << 40), "sizes too large");
    SLR_CHECK_ARG(sign == 1.0f || sign == -1.0f, "sign must be +1 or -1");
    const int blocks = (H * W + 255) / 256;
    const size_t n = (size_t)B * 2 * H * W;
    hipLaunchKernelGGL(slr::zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_motion, n);
    hipLaunchKernelGGL(slr::euler_backward_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, motion, H, W, 0,
                       steps, sign, grad_disp, grad_motion);
    SLR_CHECK_LAUNCH();
    return 0;
}
