// FETCH_SIZE / WRITE_SIZE calibration for the access patterns of the splat tile kernels (the x2 on FETCH_SIZE was calibrated on
// 16-byte streaming reads; the tile kernels gather with 4-byte buffer loads whose plane offset sits in an SGPR).  Every kernel reads
// (or writes) a KNOWN number of bytes exactly once; run each under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE        (and, in passes of their own, WRITE_SIZE / TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum ...)
// and compare the counter with the byte count printed here (tools/dev/fetch_calib.sh does that and writes profiles/r5_fetch_calibration.txt).
//   stream16      16 bytes per lane, coalesced (the guide's calibrated case: FETCH_SIZE reports exactly half)
//   stream4       4 bytes per lane, coalesced global loads
//   gather4_rows  the tile kernel's staging loads: raw_buffer_load_b32, voffset = pixel * 4 (a wave = 64 consecutive pixels of an image
//                 row, rows dealt to workgroups in a scattered order), soffset = plane * H * W * 4 in an SGPR, 4 planes per round
//   store4_rows   the tile kernel's stores: raw_buffer_store_b32 of one value per output pixel and plane, same addressing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(256) void stream16(const float4 *__restrict__ in, size_t n16, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e30f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void stream4(const float *__restrict__ in, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc == 1.2345e30f) sink[0] = acc;
}
// one workgroup (512 work-items) per 8 x 64 block of pixels, block order scrambled; all C planes, 4 per round
__global__ __launch_bounds__(512) void gather4_rows(const float *__restrict__ in, int C, int H, int W, float *sink) {
    const int tiles_x = W / 64, tiles = tiles_x * (H / 8);
    const int t = (int)(((long long)blockIdx.x * 7919) % tiles);             // (7919 is prime and larger than any tile count's factors here)
    const int y = (t / tiles_x) * 8 + (threadIdx.x >> 6), x = (t % tiles_x) * 64 + (threadIdx.x & 63);
    const uint32_t hw4 = (uint32_t)(H * W) * 4u, voff = (uint32_t)(y * W + x) * 4u;
    const rsrc_t r = make_rsrc(in, (uint32_t)C * hw4);
    float acc = 0.f;
    for (int c = 0; c < C; c += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t soff = (uint32_t)min(c + u, C - 1) * hw4;
            acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        }
    }
    if (acc == 1.2345e30f) sink[0] = acc;
}
__global__ __launch_bounds__(512) void store4_rows(float *__restrict__ out, int C, int H, int W) {
    const int tiles_x = W / 64, tiles = tiles_x * (H / 8);
    const int t = (int)(((long long)blockIdx.x * 7919) % tiles);
    const int y = (t / tiles_x) * 8 + (threadIdx.x >> 6), x = (t % tiles_x) * 64 + (threadIdx.x & 63);
    const uint32_t hw4 = (uint32_t)(H * W) * 4u, voff = (uint32_t)(y * W + x) * 4u;
    const rsrc_t r = make_rsrc(out, (uint32_t)C * hw4);
    for (int c = 0; c < C; ++c)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(c + t)), r, voff, (uint32_t)c * hw4, 0);
}

int main(int argc, char **argv) {
    const int H = 768, W = 1280;
    const int C = argc > 1 ? std::atoi(argv[1]) : 100;          // 100 planes = 393 MB: larger than the 256 MiB Infinity Cache
    const int reps = argc > 2 ? std::atoi(argv[2]) : 3;
    const size_t n = (size_t)C * H * W, bytes = n * 4;
    float *in, *out, *sink;
    OK(hipMalloc(&in, bytes)); OK(hipMalloc(&out, bytes)); OK(hipMalloc(&sink, 256));
    OK(hipMemset(in, 0, bytes)); OK(hipMemset(out, 0, bytes));
    const int tiles = (W / 64) * (H / 8);
    std::printf("bytes_per_dispatch %zu (C=%d planes of %dx%d fp32)\n", bytes, C, H, W);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const float4 *)in, n / 4, sink);
        hipLaunchKernelGGL(stream4, dim3(4096), dim3(256), 0, 0, (const float *)in, n, sink);
        hipLaunchKernelGGL(gather4_rows, dim3(tiles), dim3(512), 0, 0, (const float *)in, C, H, W, sink);
        hipLaunchKernelGGL(store4_rows, dim3(tiles), dim3(512), 0, 0, out, C, H, W);
        OK(hipDeviceSynchronize());
    }
    std::printf("done\n");
    return 0;
}
