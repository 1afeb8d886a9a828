// Minimal reproducer attempt for the packed-fp32 finding: a victim kernel whose inner loop is
// ds_read_b128 -> v_pk_fma_f32 (weight broadcast) -> store, on one stream, next to an aggressor that only issues
// MFMAs (+ LDS reads) on another stream; the victim's output is compared with the one it produces alone.
//   ./pkfma_repro [trials]        build: hipcc --offload-arch=gfx950 -O3 pkfma_repro.hip -o pkfma_repro
//   (build the victim without packed ops for the control: -Xclang -target-feature -Xclang -packed-fp32-ops)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef AGG
#define AGG 0          // aggressor: 0 = MFMA loop, 1 = the same loop with scalar FMAs (control)
#endif
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---- aggressor: MFMA + LDS reads, no global stores unless sink != 0 (never) ----
__global__ __launch_bounds__(256, 2) void aggressor(float *out, int iters, int sink) {
    __shared__ h8 frag[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) {
        h8 v;
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.001f * (float)((i + j) & 63));
        frag[i] = v;
    }
    __syncthreads();
    f16v acc[8];
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const h8 a = frag[(it * 64 + lane) & 2047];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const h8 b = frag[(it * 64 + 512 * (t & 3) + lane) & 2047];
#if AGG == 0
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
#else                                                  // control: the same loop on the vector ALU only
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = __builtin_fmaf((float)a[r & 7], (float)b[r & 7], acc[t][r]);
#endif
        }
    }
    if (sink) {
        float s = 0.0f;
        for (int t = 0; t < 8; ++t)
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

// ---- victim: the gather of the splat tile kernel reduced to its arithmetic ----
// 512 work-items; LDS holds 1536 staged float4 (4 planes of an entry) + per-pixel record lists; every work-item
// walks NREC records: ds_read_b64 (entry, weight) -> ds_read_b128 value -> 4 FMAs with the broadcast weight.
constexpr int T = 512, ENT = 1536, NREC = 6, CHUNKS = 16;
__global__ __launch_bounds__(512) void victim(const float *__restrict__ planes, const uint2 *__restrict__ recs,
                                              float *__restrict__ out, int HW) {
    __shared__ float4 val[ENT + 1];
    __shared__ uint2 rec[T * NREC];
    const int tid = threadIdx.x;
    const size_t tile = blockIdx.x;
    for (int k = 0; k < NREC; ++k) rec[tid * NREC + k] = recs[(tile * T + tid) * NREC + k];
    if (tid == 0) val[ENT] = make_float4(0.f, 0.f, 0.f, 0.f);
    float w[NREC];
    uint32_t e[NREC];
    __syncthreads();
    for (int k = 0; k < NREC; ++k) { const uint2 q = rec[tid * NREC + k]; e[k] = q.x; w[k] = __uint_as_float(q.y); }
    for (int c = 0; c < CHUNKS; ++c) {
        for (int j = 0; j < 3; ++j) {                  // stage: entry tid + j*T, 4 planes
            const size_t src = (tile * ENT + tid + j * T) % (size_t)HW;
            val[tid + j * T] = make_float4(planes[(size_t)(4 * c + 0) * HW + src], planes[(size_t)(4 * c + 1) * HW + src],
                                           planes[(size_t)(4 * c + 2) * HW + src], planes[(size_t)(4 * c + 3) * HW + src]);
        }
        __syncthreads();
        f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};          // the pairs the compiler forms in the real kernel
        float4 v[NREC];
#pragma unroll
        for (int k = 0; k < NREC; ++k) v[k] = val[e[k]];
        // VARIANT 0: the operand forms of the real kernel -- weights of two records share a register pair and are
        //            broadcast with op_sel (low register: op_sel_hi:[1,0,1]; high register: op_sel:[0,1,0])
        // VARIANT 1: only the low-register broadcast; 2: only the high-register broadcast;
        // VARIANT 3: no op_sel at all ({w, w} pairs, plain v_pk_fma_f32); 4: scalar v_fma_f32 (control)
#pragma unroll
        for (int k = 0; k < NREC; k += 2) {
            const f2 lo0 = {v[k].x, v[k].y}, hi0 = {v[k].z, v[k].w};
            const f2 lo1 = {v[k + 1].x, v[k + 1].y}, hi1 = {v[k + 1].z, v[k + 1].w};
#if VARIANT == 0
            const f2 wp = {w[k], w[k + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a01) : "v"(lo0), "v"(wp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a23) : "v"(hi0), "v"(wp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a01) : "v"(lo1), "v"(wp));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a23) : "v"(hi1), "v"(wp));
#elif VARIANT == 1
            const f2 wa = {w[k], 0.0f}, wb = {w[k + 1], 0.0f};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a01) : "v"(lo0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a23) : "v"(hi0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a01) : "v"(lo1), "v"(wb));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a23) : "v"(hi1), "v"(wb));
#elif VARIANT == 2
            const f2 wa = {0.0f, w[k]}, wb = {0.0f, w[k + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a01) : "v"(lo0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a23) : "v"(hi0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a01) : "v"(lo1), "v"(wb));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a23) : "v"(hi1), "v"(wb));
#elif VARIANT == 3
            const f2 wa = {w[k], w[k]}, wb = {w[k + 1], w[k + 1]};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(lo0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(hi0), "v"(wa));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a01) : "v"(lo1), "v"(wb));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a23) : "v"(hi1), "v"(wb));
#else
            a01.x = __builtin_fmaf(lo0.x, w[k], a01.x); a01.y = __builtin_fmaf(lo0.y, w[k], a01.y);
            a23.x = __builtin_fmaf(hi0.x, w[k], a23.x); a23.y = __builtin_fmaf(hi0.y, w[k], a23.y);
            a01.x = __builtin_fmaf(lo1.x, w[k + 1], a01.x); a01.y = __builtin_fmaf(lo1.y, w[k + 1], a01.y);
            a23.x = __builtin_fmaf(hi1.x, w[k + 1], a23.x); a23.y = __builtin_fmaf(hi1.y, w[k + 1], a23.y);
#endif
        }
        const float acc[4] = {a01.x, a01.y, a23.x, a23.y};
        for (int u = 0; u < 4; ++u) out[((size_t)(4 * c + u) * gridDim.x + tile) * T + tid] = acc[u];
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const int trials = argc > 1 ? std::atoi(argv[1]) : 200;
    const int tiles = 1920, HW = 768 * 1280, planes_n = 4 * CHUNKS;
    std::vector<float> h_planes((size_t)planes_n * HW);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (auto &x : h_planes) x = (float)(int)(rnd() >> 8) / 8388608.0f - 1.0f;
    std::vector<uint2> h_recs((size_t)tiles * T * NREC);
    for (auto &q : h_recs) {
        const uint32_t r = rnd();
        const float w = (float)(rnd() >> 8) / 16777216.0f;
        q.x = (r % 7 == 0) ? ENT : (r >> 3) % ENT;           // some records point at the all-zero entry
        std::memcpy(&q.y, &w, 4);
    }
    float *planes, *out, *ref, *agg;
    uint2 *recs;
    const size_t out_n = (size_t)planes_n * tiles * T;
    OK(hipMalloc(&planes, h_planes.size() * 4));
    OK(hipMalloc(&recs, h_recs.size() * 8));
    OK(hipMalloc(&out, out_n * 4));
    OK(hipMalloc(&ref, out_n * 4));
    OK(hipMalloc(&agg, 4096 * 256 * 4));
    OK(hipMemcpy(planes, h_planes.data(), h_planes.size() * 4, hipMemcpyHostToDevice));
    OK(hipMemcpy(recs, h_recs.data(), h_recs.size() * 8, hipMemcpyHostToDevice));
    hipStream_t sa, sv;
    OK(hipStreamCreate(&sa));
    OK(hipStreamCreate(&sv));
    hipLaunchKernelGGL(victim, dim3(tiles), dim3(T), 0, sv, planes, recs, ref, HW);
    OK(hipDeviceSynchronize());
    std::vector<float> h_ref(out_n), h_out(out_n);
    OK(hipMemcpy(h_ref.data(), ref, out_n * 4, hipMemcpyDeviceToHost));
    for (int mode = 0; mode < 2; ++mode) {                 // 0: victim alone, 1: next to the aggressor
        long wrong_runs = 0, wrong_vals = 0, lowhalf = 0;
        for (int t = 0; t < trials; ++t) {
            OK(hipMemsetAsync(out, 0xff, out_n * 4, sv));
            OK(hipStreamSynchronize(sv));
            if (mode) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 0, sa, agg, 6000, 0);
            hipLaunchKernelGGL(victim, dim3(tiles), dim3(T), 0, sv, planes, recs, out, HW);
            OK(hipDeviceSynchronize());
            OK(hipMemcpy(h_out.data(), out, out_n * 4, hipMemcpyDeviceToHost));
            long bad = 0;
            for (size_t i = 0; i < out_n; ++i)
                if (std::memcmp(&h_out[i], &h_ref[i], 4) != 0) {
                    ++bad;
                    const size_t plane = i / ((size_t)tiles * T);
                    if ((plane & 1) == 0) ++lowhalf;
                }
            wrong_vals += bad;
            wrong_runs += bad != 0;
        }
        std::printf("%s: %ld of %d runs differ from the victim's own result (%ld values, %ld of them in even planes)\n",
                    mode ? "victim next to the MFMA aggressor" : "victim alone", wrong_runs, trials, wrong_vals, lowhalf);
    }
    return 0;
}
