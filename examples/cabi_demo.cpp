// Host program on the C ABI alone (include/slr_splat.h): no Python, no torch -- hipMalloc'ed buffers, one HIP
// stream, Euler integration of a motion field followed by the summation splat and the fused softmax mode, then
// the per-clip path (all-frames Euler passes -> clip plan -> three frames of the clip synthesised by ONE launch),
// results written as raw float32 files.  tests/test_gpu_parity.py::test_c_abi_from_a_plain_host_program builds
// and runs it and compares the files with the CPU oracle.
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/cabi_demo.cpp -o cabi_demo \
//         -Lslr-sfs_amd/lib -lslrsplat -Wl,-rpath,$PWD/slr-sfs_amd/lib
//   ./cabi_demo in.f32 motion.f32 metric.f32 C H W nsteps out_prefix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slr_splat.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SLR_OK(x) do { int rc_ = (x); if (rc_ != 0) { \
    std::fprintf(stderr, "%s: rc=%d (%s)\n", #x, rc_, slr_last_error()); return 3; } } while (0)

static bool read_f32(const char *path, std::vector<float> &v) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    const size_t n = std::fread(v.data(), sizeof(float), v.size(), f);
    std::fclose(f);
    return n == v.size();
}

static bool write_f32(const char *prefix, const char *name, const float *dev, size_t n) {
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
    char path[1024];
    std::snprintf(path, sizeof path, "%s%s.f32", prefix, name);
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const size_t w = std::fwrite(h.data(), sizeof(float), n, f);
    std::fclose(f);
    return w == n;
}

int main(int argc, char **argv) {
    if (argc != 9) {
        std::fprintf(stderr, "usage: %s in.f32 motion.f32 metric.f32 C H W nsteps out_prefix\n", argv[0]);
        return 1;
    }
    const int C = std::atoi(argv[4]), H = std::atoi(argv[5]), W = std::atoi(argv[6]), nsteps = std::atoi(argv[7]);
    const char *prefix = argv[8];
    if (slr_abi_version() != SLR_ABI_VERSION) {
        std::fprintf(stderr, "library ABI %d, header %d\n", slr_abi_version(), SLR_ABI_VERSION);
        return 1;
    }
    const size_t hw = (size_t)H * W;
    std::vector<float> h_in(C * hw), h_motion(2 * hw), h_metric(hw);
    if (!read_f32(argv[1], h_in) || !read_f32(argv[2], h_motion) || !read_f32(argv[3], h_metric)) {
        std::fprintf(stderr, "cannot read the inputs\n");
        return 1;
    }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    float *in, *motion, *metric, *disp, *vis, *out_sum, *out_soft;
    HIP_OK(hipMalloc(&in, C * hw * 4));
    HIP_OK(hipMalloc(&motion, 2 * hw * 4));
    HIP_OK(hipMalloc(&metric, hw * 4));
    HIP_OK(hipMalloc(&disp, 2 * hw * 4));
    HIP_OK(hipMalloc(&vis, hw * 4));
    HIP_OK(hipMalloc(&out_sum, C * hw * 4));
    HIP_OK(hipMalloc(&out_soft, C * hw * 4));
    HIP_OK(hipMemcpyAsync(in, h_in.data(), C * hw * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(motion, h_motion.data(), 2 * hw * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(metric, h_metric.data(), hw * 4, hipMemcpyHostToDevice, st));

    // caller-owned scratch: the library allocates nothing
    const size_t ws_bytes = slr_splat_workspace_bytes(1, H, W);
    void *ws;
    HIP_OK(hipMalloc(&ws, ws_bytes));

    SLR_OK(slr_euler_integrate(motion, H, W, nsteps, 1.0f, disp, vis, st));                        // a1
    const int fe_before = slr_splat_set_front_end(2);                                                 // (the rows front end, whatever the grid size)
    SLR_OK(slr_softsplat_forward(in, disp, out_sum, 1, C, H, W, ws, ws_bytes, 0, st));              // a3 (self-contained call)
    slr_splat_set_front_end(fe_before);
    SLR_OK(slr_splat_bin(disp, 1, H, W, ws, ws_bytes, st));                                         // bins of disp, made once ...
    SLR_OK(slr_softsplat_mode_forward(in, metric, disp, out_soft, 1, C, H, W, SLR_MODE_SOFTMAX,     // a4 ... and reused
                                      ws, ws_bytes, 1, st));
    // argument errors come back as codes + message, nothing is launched
    if (slr_softsplat_forward(in, disp, out_sum, 1, C, H, W, ws, 16, 0, st) == 0) {
        std::fprintf(stderr, "a 16-byte workspace was accepted\n");
        return 4;
    }
    // ---- the frame-synthesis block of forward_flow for a clip of N frames (a6): features = the C input planes, weight
    // logits = the metric plane, frames t = 1, N/2, N-1 in one launch of the tile kernel
    const int N = nsteps + 3, NB = 3;
    const int ts[NB] = {1, N / 2, N - 1};
    float *disp_f, *disp_p, *zmax, *zscratch, *frames_out;
    HIP_OK(hipMalloc(&disp_f, (size_t)N * 2 * hw * 4));                 // forward maps of t = 0 .. N-1
    HIP_OK(hipMalloc(&disp_p, (size_t)(N + 1) * 2 * hw * 4));           // backward maps of 0 .. N steps
    HIP_OK(hipMalloc(&zmax, 4));
    HIP_OK(hipMalloc(&zscratch, 1024 * 4));
    HIP_OK(hipMalloc(&frames_out, (size_t)NB * C * hw * 4));
    SLR_OK(slr_euler_integrate_all(motion, H, W, N - 1, +1.0f, disp_f, nullptr, st));
    SLR_OK(slr_euler_integrate_all(motion, H, W, N, -1.0f, disp_p, nullptr, st));
    SLR_OK(slr_global_max(metric, hw, zmax, zscratch, st));                                         // Z.max()
    int h_idx_f[NB], h_idx_p[NB], *idx_f, *idx_p;
    for (int k = 0; k < NB; ++k) { h_idx_f[k] = ts[k]; h_idx_p[k] = N - ts[k]; }                   // t forward, N - t backward steps
    HIP_OK(hipMalloc(&idx_f, sizeof h_idx_f));
    HIP_OK(hipMalloc(&idx_p, sizeof h_idx_p));
    HIP_OK(hipMemcpyAsync(idx_f, h_idx_f, sizeof h_idx_f, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(idx_p, h_idx_p, sizeof h_idx_p, hipMemcpyHostToDevice, st));
    const size_t plan_bytes = slr_clip_plan_bytes(NB, H, W);
    void *plan;
    HIP_OK(hipMalloc(&plan, plan_bytes));
    SLR_OK(slr_clip_plan_build(disp_f, idx_f, disp_p, idx_p, NB, H, W, plan, plan_bytes, st));      // bins + plans of all frames
    const float *pf[NB], *pp[NB];
    float *po[NB], alpha[NB];
    int frame[NB];
    for (int k = 0; k < NB; ++k) {
        pf[k] = disp_f + (size_t)ts[k] * 2 * hw;
        pp[k] = disp_p + (size_t)(N - ts[k]) * 2 * hw;
        po[k] = frames_out + (size_t)k * C * hw;
        alpha[k] = 1.0f - (float)ts[k] / (float)N;                                                  // :860
        frame[k] = k;
    }
    SLR_OK(slr_synth_group_clip_batch(in, metric, zmax, 1, pf, pp, alpha, po, nullptr, C, H, W, 1e-8f, plan, plan_bytes,
                                      NB, frame, NB, nullptr /* totals not read back: upper-bound grids */, st));
    HIP_OK(hipStreamSynchronize(st));
    if (!write_f32(prefix, "frames", frames_out, (size_t)NB * C * hw)) {
        std::fprintf(stderr, "cannot write the outputs\n");
        return 1;
    }
    if (!write_f32(prefix, "disp", disp, 2 * hw) || !write_f32(prefix, "visible", vis, hw) ||
        !write_f32(prefix, "sum", out_sum, C * hw) || !write_f32(prefix, "softmax", out_soft, C * hw)) {
        std::fprintf(stderr, "cannot write the outputs\n");
        return 1;
    }
    std::printf("ok: C=%d H=%d W=%d nsteps=%d workspace %zu bytes, last error \"%s\"\n", C, H, W, nsteps, ws_bytes,
                slr_last_error());
    return 0;
}
