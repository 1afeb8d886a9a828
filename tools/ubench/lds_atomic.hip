// Microbenchmark: LDS atomic throughput on gfx950 (development aid).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
    __syncthreads();
    int a = threadIdx.x;
    float v = 1.0f + threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        int idx = (a + i * stride) & 4095;
        if (MODE == 0) atomicAdd(&lds[idx], v);                                   // ds_add_f32
        else if (MODE == 1) atomicAdd((unsigned *)&lds[idx], (unsigned)i);        // ds_add_u32
        else if (MODE == 2) { float t = lds[idx]; lds[idx] = t + v; }             // plain RMW (racy)
        else if (MODE == 3) atomicMax((int *)&lds[idx], i);                       // ds_max_i32
        else if (MODE == 4) lds[idx] = v;                                         // plain write
        else if (MODE == 6) atomicAdd(&((double *)lds)[idx & 2047], (double)v);       // ds_add_f64
        else if (MODE == 7) { unsigned o = atomicAdd((unsigned *)&lds[idx], 1u); a += (o & 1); }  // ds_add_rtn_u32
        else if (MODE == 8) { unsigned long long *p = &((unsigned long long *)lds)[idx & 2047]; atomicAdd(p, (unsigned long long)i); } // ds_add_u64
        else if (MODE == 5) { atomicAdd(&lds[idx], v); atomicAdd(&lds[(idx + 1) & 4095], v);
                              atomicAdd(&lds[(idx + 64) & 4095], v); atomicAdd(&lds[(idx + 65) & 4095], v); }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < 4096; i += 256) s += lds[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int stride, int mult) {
    const int blocks = 2048, iters = 2048;
    float *out;
    CK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks, 256>>>(out, iters, stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    k<MODE><<<blocks, 256>>>(out, iters, stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double waveinstr = (double)blocks * 4 * iters * mult;
    double cu_cycles = ms * 1e-3 * 2.1e9 * 256;   // assume 2.1 GHz
    printf("%-28s stride=%3d  %.3f ms  -> %.1f CU-cycles per wave-instr (%.2f T lane-ops/s)\n", name, stride, ms,
           cu_cycles / waveinstr, waveinstr * 64 / (ms * 1e-3) / 1e12);
    CK(hipFree(out));
}

int main() {
    for (int stride : {64, 1}) {
        run<0>("ds_add_f32", stride, 1);
        run<1>("ds_add_u32", stride, 1);
        run<2>("plain read+add+write", stride, 1);
        run<3>("ds_max_i32", stride, 1);
        run<4>("plain write", stride, 1);
        run<5>("4x ds_add_f32 (footprint)", stride, 4);
        run<6>("ds_add_f64", stride, 1);
        run<7>("ds_add_rtn_u32", stride, 1);
        run<8>("ds_add_u64", stride, 1);
    }
    return 0;
}
