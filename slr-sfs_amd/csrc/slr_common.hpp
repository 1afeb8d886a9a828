// slr_common.hpp -- shared host/device helpers for libslrsplat (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

#include "../../include/slr_splat.h"
#include "slr_tuning.hpp"

#define SLR_EXPORT extern "C" __attribute__((visibility("default")))

namespace slr {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define SLR_CHECK_ARG(cond, what)                                              \
    do { if (!(cond)) { slr::set_error("%s: bad argument: %s", __func__, what); return SLR_E_BADARG; } } while (0)

#define SLR_CHECK_HIP(expr)                                                    \
    do { hipError_t e_ = (expr);                                               \
         if (e_ != hipSuccess) { slr::set_error("%s: %s", __func__, hipGetErrorString(e_)); return (int)e_; } } while (0)

#define SLR_CHECK_LAUNCH() SLR_CHECK_HIP(hipGetLastError())

// More than 64 KiB of dynamic LDS needs an explicit opt-in per kernel and device: set once per (kernel, device), safe to call from
// several host threads (the flags are atomics; the attribute call itself is idempotent, a lost race only repeats it).
struct LdsOptIn { std::atomic<bool> done[64] = {}; };
inline int lds_opt_in(const void *kernel, int bytes, LdsOptIn &st) {
    int dev = 0;
    SLR_CHECK_HIP(hipGetDevice(&dev));                  // (a process may drive several GPUs)
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && st.done[dev].load(std::memory_order_acquire)) return 0;
    SLR_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (tracked) st.done[dev].store(true, std::memory_order_release);
    return 0;
}

// ---- geometry of the output tiling ------------------------------------------------------
// One workgroup owns a TILE_H x TILE_W block of OUTPUT pixels of one sample ("tile"); its bin
// lists the source pixels whose bilinear footprint touches the tile.  Long bins are cut into segments of SEG entries so that no workgroup gets more than
// ~2x the average work; a tile with several segments is finished by the combine kernel.
constexpr int TILE_W   = 64;      // one wavefront of consecutive x
constexpr int TILE_H   = SLR_TILE_H;
constexpr int TILE_PIX = TILE_W * TILE_H;
constexpr int SEG_ONE  = SLR_EPT_ONE * TILE_PIX;   // segment length, one flow per tile
constexpr int SEG_TWO  = SLR_EPT_TWO * TILE_PIX;   // segment length, forward+backward flows per tile

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---- bilinear footprint ------------------------------------------------------------------
// Restates models/softsplat.py:169-184 (target coordinate, NW corner, 4 weights).
struct Corners {
    int x0, y0;
    float w[4];        // NW, NE, SW, SE
    bool ok;           // coordinate representable (finite, |.| < 2^30)
};

// (X, Y) = the target coordinate (float)x + fx, (float)y + fy of softsplat.py:169-170
__device__ __forceinline__ Corners corners_at(float X, float Y) {
    Corners c;
    c.ok = (fabsf(X) < 1073741824.0f) && (fabsf(Y) < 1073741824.0f);
    float flx = floorf(X), fly = floorf(Y);
    c.x0 = c.ok ? (int)flx : 0;
    c.y0 = c.ok ? (int)fly : 0;
    float x1 = (float)(c.x0 + 1), y1 = (float)(c.y0 + 1), x0f = (float)c.x0, y0f = (float)c.y0;
    c.w[0] = (x1 - X) * (y1 - Y);
    c.w[1] = (X - x0f) * (y1 - Y);
    c.w[2] = (x1 - X) * (Y - y0f);
    c.w[3] = (X - x0f) * (Y - y0f);
    return c;
}

__device__ __forceinline__ Corners make_corners(float fx, float fy, int x, int y) {
    return corners_at((float)x + fx, (float)y + fy);
}

__device__ __forceinline__ bool in_image(int cx, int cy, int H, int W) {
    return (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H);
}

}  // namespace slr
