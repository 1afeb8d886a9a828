// Microbenchmark: achievable HBM bandwidth for the access shapes the splat kernels use (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int HW = 768 * 1280;
constexpr int C = 65;

// (a) flat copy, 4 B per lane, grid-stride
__global__ __launch_bounds__(256) void copy1(const float *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// (b) flat copy, 16 B per lane
__global__ __launch_bounds__(256) void copy4(const float4 *in, float4 *out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// (c) one thread per pixel, loops over planes U at a time (4 B per lane per plane), writes planes
template <int U>
__global__ __launch_bounds__(256) void planes1(const float *in, float *out) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[(size_t)(c + u) * HW + p];
#pragma unroll
        for (int u = 0; u < U; ++u) out[(size_t)(c + u) * HW + p] = v[u] * 1.0001f;
    }
}
// (d) one thread per 4 pixels (16 B per lane per plane)
template <int U>
__global__ __launch_bounds__(256) void planes4(const float4 *in, float4 *out) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW / 4) return;
    for (int c = 0; c + U <= C; c += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[(size_t)(c + u) * (HW / 4) + p];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u].x *= 1.0001f; out[(size_t)(c + u) * (HW / 4) + p] = v[u]; }
    }
}
// (e) read-only planes (sum), 4 B per lane
template <int U>
__global__ __launch_bounds__(256) void read1(const float *in, float *out) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    float s = 0;
    for (int c = 0; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[(size_t)(c + u) * HW + p];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    out[p] = s;
}
// (f) write-only planes
__global__ __launch_bounds__(256) void write1(float *out) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < C; ++c) out[(size_t)c * HW + p] = (float)c;
}
__global__ __launch_bounds__(256) void write4(float4 *out) {
    int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW / 4) return;
    for (int c = 0; c < C; ++c) out[(size_t)c * (HW / 4) + p] = make_float4(c, c, c, c);
}

// (g) tile copy: workgroup = TH x TW tile of every plane (the splat tile kernel's access shape)
template <int TH, int TW, int U>
__global__ __launch_bounds__(512) void tilecopy(const float *in, float *out) {
    constexpr int W = 1280, H = 768;
    const int tiles_x = W / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int ly = threadIdx.x / TW, lx = threadIdx.x % TW;
    const size_t p = (size_t)(ty0 + ly) * W + tx0 + lx;
    for (int c = 0; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[(size_t)(c + u) * HW + p];
#pragma unroll
        for (int u = 0; u < U; ++u) out[(size_t)(c + u) * HW + p] = v[u] * 1.0001f;
    }
}

template <typename F>
void timeit(const char *name, double bytes, F f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-34s %8.1f us   %6.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main() {
    size_t n = (size_t)C * HW;
    float *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4));
    double rw = 2.0 * n * 4, r = n * 4.0;
    int pb = (HW + 255) / 256, pb4 = (HW / 4 + 255) / 256;
    for (int g : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "copy dword grid=%d", g); timeit(nm, rw, [&] { copy1<<<g, 256>>>(in, out, n); });
        snprintf(nm, 64, "copy dwordx4 grid=%d", g); timeit(nm, rw, [&] { copy4<<<g, 256>>>((float4 *)in, (float4 *)out, n / 4); });
    }
    timeit("planes dword U=1", rw, [&] { planes1<1><<<pb, 256>>>(in, out); });
    timeit("planes dword U=4", rw, [&] { planes1<4><<<pb, 256>>>(in, out); });
    timeit("planes dword U=13", rw, [&] { planes1<13><<<pb, 256>>>(in, out); });
    timeit("planes dwordx4 U=1", rw, [&] { planes4<1><<<pb4, 256>>>((float4 *)in, (float4 *)out); });
    timeit("planes dwordx4 U=5", rw, [&] { planes4<5><<<pb4, 256>>>((float4 *)in, (float4 *)out); });
    timeit("planes dwordx4 U=13", rw, [&] { planes4<13><<<pb4, 256>>>((float4 *)in, (float4 *)out); });
    timeit("tile copy 8x64 U=8", rw, [&] { tilecopy<8, 64, 8><<<HW / 512, 512>>>(in, out); });
    timeit("tile copy 8x64 U=13", rw, [&] { tilecopy<8, 64, 13><<<HW / 512, 512>>>(in, out); });
    timeit("tile copy 4x128 U=13", rw, [&] { tilecopy<4, 128, 13><<<HW / 512, 512>>>(in, out); });
    timeit("tile copy 2x256 U=13", rw, [&] { tilecopy<2, 256, 13><<<HW / 512, 512>>>(in, out); });
    timeit("tile copy 1x512(640?) skip", rw, [&] { tilecopy<2, 256, 5><<<HW / 512, 512>>>(in, out); });
    timeit("tile copy 16x32 U=13", rw, [&] { tilecopy<16, 32, 13><<<HW / 512, 512>>>(in, out); });
    timeit("read-only dword U=13", r, [&] { read1<13><<<pb, 256>>>(in, out); });
    timeit("read-only dword U=5", r, [&] { read1<5><<<pb, 256>>>(in, out); });
    timeit("write-only dword", r, [&] { write1<<<pb, 256>>>(out); });
    timeit("write-only dwordx4", r, [&] { write4<<<pb4, 256>>>((float4 *)out); });
    timeit("hipMemsetAsync 255MB", r, [&] { CK(hipMemsetAsync(out, 0, n * 4)); });
    timeit("hipMemcpyAsync D2D", rw, [&] { CK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice)); });
    return 0;
}
