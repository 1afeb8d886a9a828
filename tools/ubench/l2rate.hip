// Microbenchmark: per-CU load throughput by access width when the data is L2- (or L1-) resident (gfx950).
// Question behind it: is a kernel that issues one 4-byte load per lane and instruction bound by
// the NUMBER of vector-memory instructions rather than by bytes?  Every workgroup re-reads its own small buffer
// (size per workgroup selectable: L1-resident 16 KiB, L2-resident 256 KiB ...) REP times with 4 / 8 / 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int U>
__global__ __launch_bounds__(256) void rd(const T *in, float *out, int per_wg_elems, int rep) {
    const T *p = in + (size_t)(blockIdx.x % 256) * per_wg_elems;   // <= 256 MiB touched; co-resident workgroups of a CU share a slice
    float s = 0;
    for (int r = 0; r < rep; ++r)
        for (int i = threadIdx.x; i + (U - 1) * 256 < per_wg_elems; i += U * 256) {
            T v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) s += ((const float *)&v[u])[0];
        }
    if (s == 12345.678f) out[0] = s;
}

template <typename T, int U>
void run(const char *name, const float *buf, float *out, int wgs, size_t per_wg_bytes, int rep) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int elems = (int)(per_wg_bytes / sizeof(T));
    rd<T, U><<<wgs, 256>>>((const T *)buf, out, elems, 2);
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < 5; ++i) {
        CK(hipEventRecord(a)); rd<T, U><<<wgs, 256>>>((const T *)buf, out, elems, rep); CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * per_wg_bytes * rep;
    printf("%-28s wgs=%5d per-wg=%7zu B  %8.1f us  %7.2f TB/s  %6.1f B/clk/CU(2.1GHz)  %6.1f clk per wave-instr/CU\n", name, wgs,
           per_wg_bytes, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 2.1e9,
           (best * 1e-3) * 2.1e9 / (bytes / 256.0 / (64.0 * sizeof(T))));
}

int main() {
    float *buf, *out;
    const size_t total = (size_t)512 << 20;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(buf, 0, total));
    for (size_t per : {(size_t)16 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
        for (int wgs : {256 * 2, 256 * 8}) {
            const int rep = (int)(((size_t)64 << 20) / per) / (wgs / 256) / 4 + 1;
            run<float, 4>("dword   U=4", buf, out, wgs, per, rep);
            run<float, 8>("dword   U=8", buf, out, wgs, per, rep);
            run<float2, 4>("dwordx2 U=4", buf, out, wgs, per, rep);
            run<float4, 4>("dwordx4 U=4", buf, out, wgs, per, rep);
        }
    }
    return 0;
}
