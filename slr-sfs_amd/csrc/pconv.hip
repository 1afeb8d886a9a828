// pconv.hip -- fused elementwise stages of the partial-convolution decoder.
//
// The reference's ResNet_Block_Pconv2 (models/layers/blocks.py:173-248) wraps every 3x3
// convolution in ~9 full-size elementwise passes: noise-BN scale, shift, ReLU, mask multiply
// before it (normalization.py:219-231, partialconv2d.py:69) and bias-subtract, ratio multiply,
// bias-add, update-mask multiply, residual add after it (partialconv2d.py:71-74, blocks.py:248).
// At 768x1280 these passes are ~30 % of the decoder time on MI355X (each moves 0.5 GB).
// Two kernels do the same arithmetic, in the same order, in one read + one write each.
// The convolutions themselves stay with MIOpen (north star: PyTorch-ROCm for the convs).
#include "slr_common.hpp"

namespace slr {

// y = relu(x*scale[c] - shift[c]) * mask      mask: [N,1,H,W], [N,C,H,W], or derived as (x != 0)
// MASK: 0 = one-channel, 1 = per-channel tensor, 2 = (x != 0), 3 = no mask (plain BN + ReLU of the
// non-partial blocks: models/layers/blocks.py:66-74)
template <int MASK, typename V>
__global__ __launch_bounds__(256) void bn_relu_mask_kernel(const V *__restrict__ x, const float *__restrict__ scale,
                                                           const float *__restrict__ shift,
                                                           const V *__restrict__ mask, V *__restrict__ y,
                                                           int C, int HWv) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float sc = scale[c], sh = shift[c];
    const size_t base = ((size_t)n * C + c) * HWv;
    const size_t mbase = MASK == 1 ? base : (size_t)n * HWv;
    constexpr int L = sizeof(V) / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HWv; i += gridDim.x * 256) {
        V v = x[base + i];
        V m;
        if (MASK < 2) m = mask[mbase + i];
        float *vf = reinterpret_cast<float *>(&v);
        const float *mf = reinterpret_cast<const float *>(&m);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const float xv = vf[k];
            const float r = fmaxf(xv * sc - sh, 0.0f);       // fused_bn :231, ReLU
            if (MASK == 3) { vf[k] = r; continue; }
            const float mk = MASK == 2 ? (xv != 0.0f ? 1.0f : 0.0f) : mf[k];
            vf[k] = r * mk;                                  // input*mask :69
        }
        y[base + i] = v;
    }
}

// Partial-convolution epilogue (partialconv2d.py:61-74) on the BIAS-FREE convolution output raw0:
//   o = (raw0 * ratio + b[c]) * um,   ratio = winsize/(umr + 1e-8) * um,   um = clamp(umr, 0, 1)
// (the reference forms raw0 + b inside the convolution and subtracts b again; skipping that
// round trip differs by <= 1 ulp of |raw0 + b| and saves a full-size bias-add pass)
//   RES:  o += res                                   (residual add, blocks.py:248)
//   NEXT: o  = relu(o*scale2[c] - shift2[c]) * um    (BN + ReLU + input*mask of the NEXT partial
//                                                     convolution, whose mask is this update mask)
template <bool RES, bool NEXT, typename V>
__global__ __launch_bounds__(256) void pconv_epilogue_kernel(const V *__restrict__ raw, const float *__restrict__ bias,
                                                             const V *__restrict__ umr, const V *__restrict__ res,
                                                             const float *__restrict__ scale2,
                                                             const float *__restrict__ shift2, V *__restrict__ out,
                                                             V *__restrict__ um_out, float mscale, float winsize,
                                                             int C, int HWv) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float b = bias[c];
    const float sc = NEXT ? scale2[c] : 1.0f, sh = NEXT ? shift2[c] : 0.0f;
    const size_t base = ((size_t)n * C + c) * HWv;
    constexpr int L = sizeof(V) / 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HWv; i += gridDim.x * 256) {
        V v = raw[base + i];
        V u = umr[(size_t)n * HWv + i];            // box filter of the mask; x mscale = conv(mask, ones) (:61)
        {
            float *uw = reinterpret_cast<float *>(&u);
#pragma unroll
            for (int k = 0; k < L; ++k) uw[k] *= mscale;
        }
        V r;
        if (RES) r = res[base + i];
        float *vf = reinterpret_cast<float *>(&v);
        const float *uf = reinterpret_cast<const float *>(&u);
        const float *rf = reinterpret_cast<const float *>(&r);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const float um = fminf(fmaxf(uf[k], 0.0f), 1.0f);                // :66
            // scalar / tensor is reciprocal(tensor) * scalar in torch (Tensor.__rtruediv__): same here
            const float ratio = (1.0f / (uf[k] + 1e-8f)) * winsize * um;     // :64,67
            float o = (vf[k] * ratio + b) * um;                              // :72-74
            if (RES) o += rf[k];                                             // blocks.py:248
            if (NEXT) o = fmaxf(o * sc - sh, 0.0f) * um;                     // blocks.py:233-236, partialconv2d.py:69
            vf[k] = o;
        }
        out[base + i] = v;
        if (um_out && c == 0) {                   // the update mask for the next layer, written once
            V m = u;
            float *mw = reinterpret_cast<float *>(&m);
#pragma unroll
            for (int k = 0; k < L; ++k) mw[k] = fminf(fmaxf(mw[k], 0.0f), 1.0f);
            um_out[(size_t)n * HWv + i] = m;
        }
    }
}

template <typename V, typename F>
static void launch_planes(F f, int N, int C, int HWv, hipStream_t st) {
    int bx = (HWv + 255) / 256;
    if (bx > 1024) bx = 1024;
    f(dim3(bx, C, N), st);
}

}  // namespace slr

using namespace slr;

SLR_EXPORT int slr_bn_relu_mask(const float *x, const float *scale, const float *shift, const float *mask,
                                int mask_channels, float *y, int N, int C, int H, int W, void *stream) {
    SLR_CHECK_ARG(x && scale && shift && y, "null pointer");
    SLR_CHECK_ARG(mask_channels == 0 || mask_channels == -1 || (mask && (mask_channels == 1 || mask_channels == C)), "mask");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535, "sizes");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const bool v4 = (HW % 4 == 0) && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)mask) & 15);
    const int mode = mask_channels == -1 ? 3 : mask_channels == 0 ? 2 : (mask_channels == 1 ? 0 : 1);
    if (mode >= 2) mask = nullptr;
#define LAUNCH(M, V, n)                                                                                   \
    launch_planes<V>([&](dim3 g, hipStream_t s) {                                                           \
        hipLaunchKernelGGL((bn_relu_mask_kernel<M, V>), g, dim3(256), 0, s, (const V *)x, scale, shift,   \
                           (const V *)mask, (V *)y, C, n); }, N, C, n, st)
    if (v4) { if (mode == 0) LAUNCH(0, float4, HW / 4); else if (mode == 1) LAUNCH(1, float4, HW / 4); else if (mode == 2) LAUNCH(2, float4, HW / 4); else LAUNCH(3, float4, HW / 4); }
    else    { if (mode == 0) LAUNCH(0, float, HW);      else if (mode == 1) LAUNCH(1, float, HW);      else if (mode == 2) LAUNCH(2, float, HW);      else LAUNCH(3, float, HW); }
#undef LAUNCH
    SLR_CHECK_LAUNCH();
    return 0;
}

SLR_EXPORT int slr_pconv_epilogue(const float *raw0, const float *bias, const float *mask_box, float mask_scale,
                                  const float *residual, const float *next_scale, const float *next_shift,
                                  float *out, float *um_out, float winsize, int N, int C, int H, int W,
                                  void *stream) {
    const float *um_raw = mask_box;
    SLR_CHECK_ARG(raw0 && bias && um_raw && out, "null pointer");
    SLR_CHECK_ARG(!next_scale == !next_shift, "next_scale / next_shift go together");
    SLR_CHECK_ARG(!(residual && next_scale), "residual and next-BN fusion are exclusive");
    SLR_CHECK_ARG(N > 0 && C > 0 && H > 0 && W > 0 && C <= 65535 && N <= 65535, "sizes");
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    const bool v4 = (HW % 4 == 0) && !(((uintptr_t)raw0 | (uintptr_t)out | (uintptr_t)um_raw | (uintptr_t)residual | (uintptr_t)um_out) & 15);
#define LAUNCH(R, X, V, n)                                                                                 \
    launch_planes<V>([&](dim3 g, hipStream_t s) {                                                            \
        hipLaunchKernelGGL((pconv_epilogue_kernel<R, X, V>), g, dim3(256), 0, s, (const V *)raw0, bias,    \
                           (const V *)um_raw, (const V *)residual, next_scale, next_shift, (V *)out,      \
                           (V *)um_out, mask_scale, winsize, C, n); }, N, C, n, st)
    if (v4) {
        if (residual) LAUNCH(true, false, float4, HW / 4);
        else if (next_scale) LAUNCH(false, true, float4, HW / 4);
        else LAUNCH(false, false, float4, HW / 4);
    } else {
        if (residual) LAUNCH(true, false, float, HW);
        else if (next_scale) LAUNCH(false, true, float, HW);
        else LAUNCH(false, false, float, HW);
    }
#undef LAUNCH
    SLR_CHECK_LAUNCH();
    return 0;
}
