/*
 * slr_splat.h -- C ABI of libslrsplat.so: MI355X (gfx950) kernels for the SLR-SFS
 * frame-synthesis hot path (Euler-integrated feature warping + softmax splatting).
 *
 * The reference has no FFI for this path: models/softsplat.py hands raw data_ptr()s to
 * cupy-JIT-compiled CUDA kernels (softsplat.py:408-416) and euler_integration is a Python
 * loop of torch ops (models/projection/euler_integration_manipulator.py:36-55).  Each entry
 * point below names the reference code it replaces (file:line under the reference root).
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every tensor is fp32, NCHW, contiguous, resident in device (HBM) memory -- the same
 *     contract the reference asserts (softsplat.py:397-402);
 *   - `stream` is a hipStream_t (NULL = default stream) that belongs to the CURRENT device;
 *     all work is enqueued on it, nothing synchronises the host, nothing is allocated;
 *   - scratch memory is caller-owned: `ws` must hold slr_splat_workspace_bytes(N,H,W) bytes, 16-byte aligned (it depends on the
 *     flow's shape only, not on the planes splatted with it), and must not be shared by calls in flight on different streams;
 *   - sizes: H < 2^24, H*W < 2^26, N*H*W < 2^29; the plane count C is free -- a sample's plane stack of 2 GiB or more (the reference
 *     indexes up to 2^31 ELEMENTS per tensor, softsplat.py:163,408-416) is rendered by several launches over plane groups;
 *   - return value 0 = success; >0 = hipError_t; <0 = SLR_E_* argument error.
 *     slr_last_error() returns a thread-local description of the last failure.
 *   - a non-finite or |.| >= 2^30 target coordinate drops all four corners (reference: UB).
 */
#ifndef SLR_SPLAT_H
#define SLR_SPLAT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLR_ABI_VERSION 10

#define SLR_E_BADARG   (-1)   /* null pointer / non-positive size / unknown enum  */
#define SLR_E_WORKSPACE (-2)  /* workspace too small or misaligned                */

/* FunctionSoftsplat strType (softsplat.py:667) */
#define SLR_MODE_SUMMATION 0
#define SLR_MODE_AVERAGE   1
#define SLR_MODE_LINEAR    2
#define SLR_MODE_SOFTMAX   3

/* normaliser handling of slr_splat_normalize / slr_synth_group */
#define SLR_NORM_ZERO_TO_ONE 0   /* norm==0 -> 1      (softsplat.py:684)                       */
#define SLR_NORM_CLAMP_EPS   1   /* max(norm, eps)    (animating_softmax_splating.py:923)      */

int         slr_abi_version(void);
const char *slr_last_error(void);

/* Measurement hook: the NEXT splat call of this thread records hipEvent_t `ev_start` right
 * before and `ev_stop` right after its tile kernel (the dominant kernel; not the plan/combine
 * helpers) on the launch stream.  One-shot; NULLs disable.  Used by bench.py for the roofline. */
void slr_splat_time_next(void *ev_start, void *ev_stop);

/* ------------------------------------------------------------------ Euler integration */

/* euler_integration(motion, nsteps) for one sample.
 * Replaces models/projection/euler_integration_manipulator.py:7-56 (return_all_frames=False).
 *   motion  [2,H,W]  ch0 = x velocity, ch1 = y velocity (px/frame); multiplied by `sign`
 *                    (+1 / -1: the models call it with flow and -flow,
 *                    animating_softmax_splating.py:847-848; an exact sign flip)
 *   disp    [2,H,W]  out; invalid pixels = max(H,W)+1 in both channels (:55)
 *   visible [H,W]    out, 1.0/0.0 (:54); may be NULL
 * Bit-exact with the reference (fp32 adds, round-half-even gather index). */
int slr_euler_integrate(const float *motion, int H, int W, int nsteps, float sign,
                        float *disp, float *visible, void *stream);

/* All frames t = 0..nmax in ONE pass: disp_all[t] == euler_integration(sign*motion, t).
 * Replaces the per-frame re-integration of forward_flow (O(N^2) steps per clip,
 * animating_softmax_splating.py:847-848); the reference's own return_all_frames=True branch
 * is broken (euler_integration_manipulator.py:31,50).
 *   disp_all [nmax+1,2,H,W] out;  vis_all [nmax+1,H,W] out, may be NULL */
int slr_euler_integrate_all(const float *motion, int H, int W, int nmax, float sign,
                            float *disp_all, float *vis_all, void *stream);

/* Gradient of slr_euler_integrate w.r.t. the motion field: what torch autograd computes through the reference's
 * differentiable loop (euler_integration_manipulator.py:36-55; the training path feeds it the motion regressor's
 * output, animating_softmax_splating.py:515-580).  Each step of a still-valid pixel's path adds the pixel's
 * displacement gradient to the cell it gathered from (:37-38); pixels that left the image contribute nothing
 * (:45-46,55).
 *   grad_disp [2,H,W] in;  grad_motion [2,H,W] out (zeroed here, then accumulated with fp32 atomics) */
int slr_euler_backward(const float *motion, int H, int W, int nsteps, float sign, const float *grad_disp,
                       float *grad_motion, void *stream);

/* EulerIntegration(opt).forward(motion, destination_frame) -- the batch form the training step calls
 * (models/projection/euler_integration_manipulator.py:58-71, used at animating_softmax_splating.py:579-580): every sample
 * b integrated for its own steps[b] in ONE launch; the step counts stay on the device (the reference loops over b in
 * Python and reads destination_frame[b] on the host).
 *   motion [B,2,H,W];  steps [B] int64 ON THE DEVICE (what `index.long()` arithmetic yields; <= 0: no step, :36)
 *   disp [B,2,H,W] out;  visible [B,H,W] out, may be NULL.  Per sample bit-exact with slr_euler_integrate. */
int slr_euler_integrate_batch(const float *motion, const long long *steps, int B, int H, int W, float sign,
                              float *disp, float *visible, void *stream);

/* Gradient of slr_euler_integrate_batch w.r.t. the motion fields (per sample: slr_euler_backward).
 *   grad_disp [B,2,H,W] in;  grad_motion [B,2,H,W] out (zeroed here) */
int slr_euler_backward_batch(const float *motion, const long long *steps, int B, int H, int W, float sign,
                             const float *grad_disp, float *grad_motion, void *stream);

/* ------------------------------------------------------------------ splat: workspace, binning, front ends */

/* flags of the `prebinned` argument of the one-flow calls (0: a self-contained call on a workspace nothing is known about) */
#define SLR_WS_PREBINNED 1   /* `ws` was filled by slr_splat_bin / slr_splat_bin_pair with this flow */
#define SLR_WS_CLEAN     2   /* `ws` was zeroed by slr_splat_workspace_init and has only been used through this library since: the call
                                skips the kernel that zeroes the binning counters (the binning leaves them zero again) */

/* Bytes of scratch one flow field [N,2,H,W] needs: per-tile row-segment lists (256 records of 8 bytes per 8x64 output tile), work
 * plans, destination boxes, and for the scan front end's sink launch an entry array + slabs of partial sums -- 21 MB at 768x1280, 81 MB
 * on grids of up to 1024 tiles (where the scan front end runs by default: 16 MB of entries + 64 MB of slabs).  slr_splat_workspace_init
 * zeroes the counters of a fresh workspace (see SLR_WS_CLEAN). */
size_t slr_splat_workspace_bytes(int N, int H, int W);
int slr_splat_workspace_init(void *ws, size_t ws_bytes, int N, int H, int W, void *stream);

/* Sort `flow` [N,2,H,W] for splatting, inside `ws`: every 64-pixel row segment of the flow is appended to the few 8x64 OUTPUT
 * tiles its bilinear footprints touch (one 64-bit atomic per (segment, tile) = list slot + the tile's exact entry count), and the
 * work plan is written by the same launch (heavy tiles first; a tile of more than 1024 entries cut into ranges of its output
 * columns).  Depends on the flow only -- every tensor splatted with this flow reuses it (prebinned = SLR_WS_PREBINNED).
 * (No reference counterpart: the reference scatters with global atomics, softsplat.py:186-199; here each workgroup owns an output
 * tile -- or a range of its columns -- and gathers exactly the sources that land in it.) */
int slr_splat_bin(const float *flow, int N, int H, int W, void *ws, size_t ws_bytes, void *stream);

/* slr_splat_bin for two flow fields of the same shape (the forward and the backward displacement map of a frame). */
int slr_splat_bin_pair(const float *flow_a, const float *flow_b, int N, int H, int W,
                       void *ws_a, void *ws_b, size_t ws_bytes, void *stream);

/* Front end of the self-contained one-flow calls below (slr_softsplat_forward, slr_softsplat_mode_forward, slr_maxsplat_forward,
 * slr_max_warp_norm without SLR_WS_PREBINNED).  Two exact front ends find an output tile's source pixels:
 *   rows : slr_splat_bin's binning + plan, then the tile kernel walks exactly the listed rows of the flow; pieces of heavy tiles
 *          run in parallel, each owning its output columns (no partial tiles, no combine).  3 launches (4 on a workspace that is not
 *          SLR_WS_CLEAN): rows + plan, tile kernel, and a normally empty pass-by-pass launch for pieces that still hold more than
 *          1024 entries;
 *   scan : one kernel writes the destination box of every 8x64 block of source pixels, then every output tile's workgroup lists the
 *          rows of the blocks whose box touches it and walks them with the same code (no plan: nothing to wait for on grids that fit
 *          the chip in one or two rounds).  3 launches: boxes, tile kernel, and a normally empty SINK launch for tiles of more than
 *          1024 entries (pile-ups of a contracting flow): their workgroups write the tile's entries out once, the sink launch
 *          renders them as tasks of exactly 1024 entries, up to 16 workgroups x 8 channel groups per tile at once, each into a slab
 *          of its own; the tile's last workgroup adds the slabs up in slot order (reproducible, no float atomics) and normalises.
 * A call takes `scan` when its grid has at most `max_tiles` output tiles (N * ceil(H/8) * ceil(W/64); default 1024, 0 = never,
 * INT_MAX = always) and `rows` above; slr_splat_set_front_end(1 | 2) forces scan | rows, anything else = automatic.  Process-wide;
 * both return the previous value (-1 = automatic).  Both are exact (floating-point summation order differs: results agree to
 * rounding, ~1e-6 relative). */
int slr_splat_set_scan_max_tiles(int max_tiles);
int slr_splat_set_front_end(int front_end);
/* Tuning of the scan front end (process-wide, 0 = the built-in choice by grid size): column pieces per output tile in the first launch
 * (1, 2, 4 or 8) x channel groups per piece; piece slots (default 33: the emergency slabs of pieces that find the slab pool empty; the sink
 * launch's task list takes min(32, slots) pieces per round) x channel groups (default 8) of the sink launch. */
void slr_splat_set_scan_shape(int pieces, int groups, int defer_wg, int defer_groups);

/* ------------------------------------------------------------------ splat: forward */

/* _FunctionSoftsplat.forward: summation splat.
 * Replaces kernel_Softsplat_updateOutput + its launcher, softsplat.py:157-202, 390-424.
 *   in [N,C,H,W], flow [N,2,H,W] -> out [N,C,H,W] (every element written; no pre-zeroing)
 * prebinned: 0 or SLR_WS_CLEAN = a self-contained call (front end: slr_splat_set_front_end / slr_splat_set_scan_max_tiles);
 * SLR_WS_PREBINNED: `ws` was filled by slr_splat_bin / slr_splat_bin_pair with this flow (one binning shared by several tensors
 * splatted with the same flow). */
int slr_softsplat_forward(const float *in, const float *flow, float *out,
                          int N, int C, int H, int W,
                          void *ws, size_t ws_bytes, int prebinned, void *stream);

/* FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType) fused: weighting, splat and
 * normalisation in one pass.  Replaces softsplat.py:665-690.
 *   mode = SLR_MODE_*; metric [N,1,H,W] (ignored for SUMMATION/AVERAGE, may be NULL)
 *   out [N,C,H,W] = splat(in*m) / splat(m) with splat(m)==0 -> 1   (m = 1 | metric | exp(metric)) */
int slr_softsplat_mode_forward(const float *in, const float *metric, const float *flow, float *out,
                               int N, int C, int H, int W, int mode,
                               void *ws, size_t ws_bytes, int prebinned, void *stream);

/* Normalisation of a raw accumulation whose LAST channel is the normaliser:
 * accum [N,C+1,H,W] -> out [N,C,H,W].  Replaces softsplat.py:681-686 (ZERO_TO_ONE) and
 * animating_softmax_splating.py:923-924 (CLAMP_EPS, eps = 1e-8). */
int slr_splat_normalize(const float *accum, float *out, int N, int C, int H, int W,
                        int norm_mode, float eps, void *stream);

/* The frame-synthesis block of forward_flow for one group of planes sharing a weight plane:
 *   S   = splat(values*w*a, disp_f) + splat(values*w*(1-a), disp_p)
 *   nrm = splat(w*a, disp_f)        + splat(w*(1-a), disp_p)
 *   out = S / max(nrm, eps)                  (exact 0 where nothing lands)
 * with w = exp(wlogit - *wmax) if wmax != NULL else exp(wlogit) if exp_weights else wlogit.
 * Replaces animating_softmax_splating.py:849-862, 884-924 (values = start_fs, wlogit = Z,
 * a = 1 - t/N) and ..._2layers_alpha_seperate.py:950-1045 (second group: values = alpha_fluid,
 * wlogit = CompositeFluidAlpha_I0).  One sample (the models run bs = 1 at inference).
 *   values [C,H,W]; wlogit [H,W]; wmax: device pointer to 1 float or NULL
 *   disp_f, disp_p [2,H,W]; ws_f, ws_p: workspaces ALREADY binned with disp_f / disp_p (slr_splat_bin_pair); the call sorts copies
 *   of their lists, writes a two-flow plan into ws_f and runs the fused kernel of the clip path on that one frame
 *   out [C,H,W]; norm_out [H,W] or NULL (the clamped normaliser, for alpha_fluid_mask :1039) */
int slr_synth_group(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                    const float *disp_f, const float *disp_p, float alpha,
                    float *out, float *norm_out, int C, int H, int W, float eps,
                    void *ws_f, void *ws_p, size_t ws_bytes, void *stream);

/* ---- clip plans: all frames of a clip binned and planned by one set of launches ----
 * forward_flow integrates and splats per frame (animating_softmax_splating.py:847-848,884-921); every displacement map of a clip
 * exists before its first frame (slr_euler_integrate_all), so the 2 x n maps of n frames are binned by ONE launch (row segments
 * per tile, exact column-octant histograms), their lists sorted by one, and one workgroup per frame writes the frame's work plan
 * (tiles of more than 1536 entries of the two directions together cut into column pieces): 4 launches per clip.
 * Frame i uses disp_f[idx_f[i]] and disp_p[idx_p[i]] (idx_*: DEVICE int arrays; for frame t of an N-frame clip idx_f = t,
 * idx_p = N - t).  At most 16384 frames per plan.
 * slr_clip_plan_totals: where the per-frame totals sit inside the plan buffer (stride_words uint32 per frame: [0] work items) --
 * read them back once per clip to pass exact grids to the synthesis calls (n_items; -1 or a NULL array = unknown: upper-bound grids,
 * surplus workgroups exit at once).
 * The synthesis calls WRITE into the plan (a frame's list of pieces that turned out to hold more than a segment, emptied again by the
 * same call): one synthesis call at a time per plan -- not from two streams at once -- and no frame twice in one batch. */
size_t slr_clip_plan_bytes(int nframes, int H, int W);
int slr_clip_plan_totals(int nframes, int H, int W, size_t *offset_bytes, int *stride_words);
int slr_clip_plan_build(const float *disp_f, const int *idx_f, const float *disp_p, const int *idx_p, int nframes,
                        int H, int W, void *plan, size_t plan_bytes, void *stream);
/* slr_synth_group for nb <= 16 frames of a built clip plan in ONE launch of the fused tile kernel (+ one normally empty launch for
 * pieces of more than a segment): consecutive kernels of a stream do not overlap, and the last round of a frame's ~2100 work items
 * runs on a half-empty chip; the frames' block groups are interleaved so that the same tile of consecutive frames runs side by
 * side on one XCD and shares its L2.  Per frame of work at 768x1280: 239 us with 1 frame per launch, 158 with 8, 151 with 16.
 * Arrays of nb entries: disp_f / disp_p / out / norm_out (device pointers per frame; norm_out may be NULL), alpha,
 * frame (index into the plan, every frame at most once); n_items = nb work-item counts (slr_clip_plan_totals) or NULL (all unknown). */
/* (ABI 8) exp_weights of the three clip calls below is a set of flags: bit 0 = exp weights (as before), SLR_SYNTH_VALUES_B4 = `values` is
 * plane-blocked by 4 in memory, [C/4][H][W][4] (C % 4 == 0, the stack below 2 GiB), as slr_pack_planes4 writes it: the 4 planes of a chunk
 * are then ONE 16-byte load per source pixel instead of four 4-byte loads.  The outputs stay [C,H,W]; same arithmetic in the same order. */
#define SLR_SYNTH_VALUES_B4 2
/* in [N,C,H,W] -> out [N,C/4,H,W,4] (C % 4 == 0; out 16-byte aligned, not in place): once per clip for its feature planes. */
int slr_pack_planes4(const float *in, float *out, int N, int C, int H, int W, void *stream);
int slr_synth_group_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                               const float *const *disp_f, const float *const *disp_p, const float *alpha,
                               float *const *out, float *const *norm_out, int C, int H, int W, float eps,
                               void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                               const int *n_items, void *stream);

/* slr_synth_group_clip_batch with a SECOND WEIGHT GROUP splatted by the same launch: ONE more value plane with its own
 * weight plane -- the alpha plane of the 2-layer model, which the reference splats with CompositeFluidAlpha_I0 as weights
 * next to the 64 feature planes weighted by Z (..._2layers_alpha_seperate.py:963-1045).  The two groups share the flow, so
 * they share the per-tile records: the records keep the pure bilinear weights, the first group's weight multiplies its values
 * when they are staged, and one extra chunk (the first group's weight | values2 * w2 | w2 per source pixel) yields both
 * normalisers and the second group's sum -- instead of a second launch that rebuilds every tile's records for one plane
 * (41 us per 768x1280 frame).
 *   values2 [H,W], wlogit2 [H,W]: w2 = exp(wlogit2) if exp_weights2 else wlogit2;  out2[k] [H,W] per frame:
 *   out2 = (splat(values2*w2*a, disp_f) + splat(values2*w2*(1-a), disp_p)) / max(same for w2, eps). */
int slr_synth_two_groups_clip_batch(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                                    const float *values2, const float *wlogit2, int exp_weights2,
                                    const float *const *disp_f, const float *const *disp_p, const float *alpha,
                                    float *const *out, float *const *out2, int C, int H, int W, float eps,
                                    void *plan, size_t plan_bytes, int nframes, const int *frame, int nb,
                                    const int *n_items, void *stream);
/* slr_synth_group for frame `frame` of a built clip plan (disp_f / disp_p: that frame's two maps). */
int slr_synth_group_clip(const float *values, const float *wlogit, const float *wmax, int exp_weights,
                         const float *disp_f, const float *disp_p, float alpha, float *out, float *norm_out,
                         int C, int H, int W, float eps, void *plan, size_t plan_bytes, int nframes, int frame,
                         int n_items, void *stream);

/* Global max of a tensor (Z.max(), animating_softmax_splating.py:855) -> result[0].
 * scratch: 1024 floats of device memory. */
int slr_global_max(const float *x, size_t n, float *result, float *scratch, void *stream);

/* ------------------------------------------------------------------ splat: backward */

/* _FunctionSoftsplat.backward.  Replaces kernel_Softsplat_updateGradInput / updateGradFlow
 * and their launcher, softsplat.py:204-255, 257-326, 427-478.
 *   grad_in [N,C,H,W] and/or grad_flow [N,2,H,W]; either may be NULL (needs_input_grad).  With both requested ONE kernel
 *   gathers grad_out once for both; each result is bit-identical to the call that asks for it alone. */
int slr_softsplat_backward(const float *in, const float *flow, const float *grad_out,
                           float *grad_in, float *grad_flow,
                           int N, int C, int H, int W, void *stream);

/* The same with scratch for channel groups (round 6).  The kernel's channels are dealt to 2-4 workgroups per tile: 2-4 on grids
 * smaller than the chip -- the reference TRAINS at 256 x 256, batch 2 per GPU (train_animating_scripts/train_baseline2_pconv.sh:14):
 * 256 source tiles, one workgroup per CU walking all 65 channels --, 2 on larger ones (halves the life of the blocks a flow's sinks
 * make slow: 65 x 768 x 1280 at Euler t=59 230 -> 210 us).  gradInput is per channel (bit-identical); group 0 writes gradFlow itself,
 * the other groups' partial sums go to `ws` and a second small launch adds them on in group order (reproducible; the grouping of the
 * channel sum differs from the one-group kernel by rounding).  slr_softsplat_backward_ws_bytes: bytes of `ws` this shape wants
 * ((groups - 1) * N * 2 * H * W * 4; 0: fewer than 16 channels); a NULL / short `ws` = one group when grad_flow is asked for.
 * slr_softsplat_backward is this call without scratch. */
size_t slr_softsplat_backward_ws_bytes(int N, int C, int H, int W);
int slr_softsplat_backward_ws(const float *in, const float *flow, const float *grad_out, float *grad_in,
                              float *grad_flow, int N, int C, int H, int W, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------ maximum-splat family */

/* _FunctionMaximumsplat.forward: out[corner] = max(init, max over sources of in*w).
 * Replaces kernel_Maximumsplat_updateOutput, softsplat.py:12-82, 482-518 (init 0.0, :497). */
int slr_maxsplat_forward(const float *in, const float *flow, float *out, float init,
                         int N, int C, int H, int W,
                         void *ws, size_t ws_bytes, int prebinned, void *stream);

/* _FunctionMaximumWarpNormsplat: max-splat seeded with -1000, then per source pixel the max
 * over its in-bounds corners and itself.  Replaces softsplat.py:84-155, 576-624.
 *   scratch [N,C,H,W] receives the intermediate max-warped tensor. */
int slr_max_warp_norm(const float *in, const float *flow, float *scratch, float *out,
                      int N, int C, int H, int W,
                      void *ws, size_t ws_bytes, int prebinned, void *stream);

/* ------------------------------------------------------------------ decoder elementwise stages (8 f3) */

/* y = relu(x*scale[c] - shift[c]) * mask: eval-mode noise-BN (zero noise), ReLU and the
 * input*mask of the following partial convolution in one pass.  Replaces
 * models/layers/normalization.py:219-231 + blocks.py:229-231 + partialconv2d.py:69.
 *   mask_channels = 1: mask [N,1,H,W];  = C: mask [N,C,H,W];  = 0: mask = (x != 0)
 *   (models/networks/architectures.py:369);  = -1: no mask (plain BN + ReLU of the encoder /
 *   background blocks, blocks.py:66-74).  `mask` is ignored for 0 and -1. */
int slr_bn_relu_mask(const float *x, const float *scale, const float *shift, const float *mask,
                     int mask_channels, float *y, int N, int C, int H, int W, void *stream);

/* Partial-convolution epilogue on the bias-free convolution output raw0:
 *   um_raw = mask_box*mask_scale  (= conv(mask, ones[out,in,k,k]), partialconv2d.py:61: the k x k box
 *            filter of a channel-uniform mask times Cin, or of the channel sum of the mask times 1)
 *   o = (raw0*ratio + b)*um,  um = clamp(um_raw,0,1),  ratio = winsize/(um_raw + 1e-8)*um
 * then optionally  o += residual            (blocks.py:248)
 * or               o  = relu(o*next_scale - next_shift)*um   (BN + ReLU + input*mask of the next
 *                       partial convolution of the block, blocks.py:233-236 / partialconv2d.py:69).
 * Replaces models/layers/partialconv2d.py:64-74.
 *   raw0 [N,C,H,W]; mask_box [N,1,H,W]; residual [N,C,H,W] or NULL; next_scale/next_shift [C] or
 *   NULL (exclusive with residual); um_out [N,1,H,W] or NULL receives um; winsize = Cin*k*k. */
int slr_pconv_epilogue(const float *raw0, const float *bias, const float *mask_box, float mask_scale,
                       const float *residual, const float *next_scale, const float *next_shift,
                       float *out, float *um_out, float winsize, int N, int C, int H, int W, void *stream);

/* The split-f16 convolution kernels below represent an activation x as two f16 halves of x * xscale (`xscale` of the
 * forward calls: a power of two in (0, 64]; 64 = the default): exact-domain for |x| < 65472 / xscale -- 1023 at 64
 * (post-BN activations are O(1 .. 10^2)), 65472 at 1 (the low halves of values below 2^-3 are then f16 subnormals:
 * absolute error <= 2^-25 per value).  A larger activation is CLAMPED there (no inf / NaN), which makes the frame wrong
 * rather than inexact -- the reference's fp32 convolution has no such limit -- so every wave that had to clamp adds to
 * a counter.  The counter is ONE PER DEVICE (shared by all streams and host threads using that device).
 *   slr_conv_saturation_count: *count = the counter of the current device after everything enqueued on `stream` so far
 *       (synchronises that stream with the host); reset != 0 zeroes it.
 *   slr_conv_saturation_record: asynchronous -- the current value is copied into *host_slot (PINNED host memory) in
 *       stream order; read it after synchronising.  Differences of consecutive records tell which piece of work clamped.
 * The Python networks react to a non-zero count (smaller xscale, then fp32 convolutions, or an exception: nets.py). */
int slr_conv_saturation_count(unsigned long long *count, int reset, void *stream);
int slr_conv_saturation_record(unsigned *host_slot, void *stream);

/* ------------------------------------------------------------------ decoder convolution on the matrix cores (8 f3) */

/* 3x3 / stride 1 / zero-pad 1 convolution, fp32 in / fp32 out: implicit GEMM on
 * v_mfma_f32_32x32x16_f16 with split operands (x = hi + lo in f16, three MFMAs per product, fp32
 * accumulation; csrc/conv.hip): fp32-class accuracy (2-5e-6 abs on outputs of magnitude 4, like
 * MIOpen's fp32 Winograd) at several times the rate of the fp32 matrix pipe.  Any Cin / Cout
 * (channels are zero-padded to multiples of 16 / 32 inside the weight buffer).
 *   Weights are prepared once per layer with slr_conv3x3_split_weights into a buffer of
 *   slr_conv3x3_weight_bytes(Cout, Cin) bytes; `wscale` is a power of two that brings max|w|
 *   near 2^12 (f16 range) and must be passed unchanged to the forward calls. */
/* `layout` flags of the forward calls: the activation BETWEEN two of these kernels may be kept channel-blocked,
 * [N, C/8, H, W, 8] instead of [N, C, H, W] (C % 8 == 0): the consumer then stages an item's 8 channels with two
 * 16-byte loads instead of eight 4-byte loads, the producer stores 16 bytes per lane without a transpose.
 * Same values, same arithmetic; only the memory order of that one tensor differs. */
#define SLR_CONV_IN_B8  1      /* `in` / `x` is channel-blocked */
#define SLR_CONV_OUT_B8 2      /* `out` is written channel-blocked */
#define SLR_CONV_RES_B8 4      /* `residual` is channel-blocked (only together with SLR_CONV_OUT_B8) */
/* The fp32 rung (ABI 6): with SLR_CONV_F32 in `layout` the same entry points run the convolution on v_mfma_f32_32x32x2_f32 --
 * fp32 operands, fp32 products, fp32 accumulation: the arithmetic of the reference's own convolutions
 * (models/layers/partialconv2d.py:61-74, models/layers/blocks.py:173-248), no pre-scale, no clamp, no limit on the magnitude of the
 * activations -- at the rate of the fp32 matrix pipe (157 TFLOP/s, 1/16 of the f16 rate).  `wsplit` must then come from
 * slr_conv3x3_f32_weights / slr_conv1x1_f32_weights (same byte counts as the split-f16 buffers), wscale = xscale = 1. */
#define SLR_CONV_F32    8
/* Together with SLR_CONV_F32 on the 3x3 entry points (Cout > 4): the convolution as Winograd F(2x2, 3x3) on the same fp32 matrix
 * instructions -- 16 multiplications per (input channel, output channel, 2x2 output tile) instead of 36, the transforms are additions
 * (csrc/conv_wino.hpp).  fp32 operands, products and accumulation as on the plain fp32 rung; the rounding error of the transform
 * domain is within a small factor of the direct fp32 convolution's (tests/test_gpu_conv_f32.py: both against fp64).  `wsplit` must then
 * come from slr_conv3x3_wino_weights (slr_conv3x3_wino_weight_bytes: 16 transformed values per weight instead of 9). */
#define SLR_CONV_WINO   16
#define SLR_CONV_SKIP_B8 32    /* slr_conv3x3_forward_skip / slr_pconv3x3_forward_skip: `skip_in` is channel-blocked */
#define SLR_CONV_POOL_OUT 64   /* ... and `out` is avgpool3x3s2 of the result, [N,Cout,(H-1)/2+1,(W-1)/2+1] channel-blocked (needs pool_ws) */
#define SLR_CONV_UP_OUT 128    /* ... or `out` is the x2 bilinear up-sampling of the result, [N,Cout,2H,2W] channel-blocked (needs pool_ws) */
/* Cout <= 4 (the 128 -> 3 end of the decoders): the 3x3 entry points run a kernel of their own on EITHER rung -- fp32 FMAs on the vector
 * ALUs (csrc/conv_few.hpp; the narrowest matrix-core tile would compute 32 channels for 3), i.e. the reference's arithmetic: both
 * weight-preparation calls then write plain fp32 weights into the buffer, wscale / xscale are accepted and unused, nothing saturates. */

size_t slr_conv3x3_weight_bytes(int Cout, int Cin);
int slr_conv3x3_split_weights(const float *w /* [Cout,Cin,3,3] */, void *wsplit, int Cout, int Cin,
                              float wscale, void *stream);
int slr_conv3x3_f32_weights(const float *w /* [Cout,Cin,3,3] */, void *wfrag /* slr_conv3x3_weight_bytes */, int Cout, int Cin,
                            void *stream);
size_t slr_conv3x3_wino_weight_bytes(int Cout, int Cin);
int slr_conv3x3_wino_weights(const float *w /* [Cout,Cin,3,3] */, void *wfrag /* slr_conv3x3_wino_weight_bytes */, int Cout, int Cin,
                             void *stream);

/* out = conv3x3(pre(in)) + bias + residual, pre(x) = relu(x*pre_scale[c] - pre_shift[c]) when pre_scale
 * is given (eval-mode noise-BN + ReLU in front of the convolution, models/layers/blocks.py:66-74 +
 * normalization.py:219-231), identity otherwise.  bias [Cout] or NULL; residual [N,Cout,H,W] or NULL
 * (the x_a + x_b of ResNet_Block, blocks.py:87). */
int slr_conv3x3_forward(const float *in, const void *wsplit, const float *bias, const float *residual, float *out,
                        int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                        const float *pre_scale, const float *pre_shift, int layout, void *stream);

/* One partial convolution of ResNet_Block_Pconv2 in a single kernel
 * (models/layers/partialconv2d.py:41-81 with blocks.py:229-239,248):
 *   xin = relu(x*pre_scale - pre_shift) * mask     (prologue; skipped when pre_scale is NULL: x is
 *         then the already activated and masked input, i.e. the output of a previous call with
 *         next_scale / next_shift)
 *         mask [N,1,H,W]: channel-uniform mask;  mask = NULL: the per-element mask (x != 0) of
 *         models/networks/architectures.py:369 (needs pre_scale: x must be the raw input)
 *   raw0 = conv3x3(xin)                             (bias-free)
 *   um_raw = conv(mask, ones[Cout,Cin,3,3]) (:61) = box3x3(mask)*Cin, resp. box3x3(sum_c (x != 0));
 *            computed inside the kernel from the mask plane of the block's halo (exact integers)
 *   out  = epilogue(raw0) exactly as slr_pconv_epilogue: (raw0*ratio + b)*um, then `+ residual`
 *          or relu(.*next_scale - next_shift)*um;  um = clamp(um_raw, 0, 1) -> um_out.
 * Same operations in the same order as slr_bn_relu_mask -> convolution -> slr_pconv_epilogue. */
int slr_pconv3x3_forward(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                         const void *wsplit, float wscale, float xscale, const float *bias, const float *residual,
                         const float *next_scale, const float *next_shift, float *out, float *um_out,
                         int N, int Cin, int Cout, int H, int W, int layout, void *stream);

/* ABI 10.  The same two convolutions with the residual block's 1x1 skip branch INSIDE the kernel (models/layers/blocks.py:83-87 and
 * :243-248: x_a + x_b with x_b = conv1x1(block input)):
 *   out = [3x3 convolution with its whole epilogue, as above, without residual / next-BN] + conv1x1(skip_in) (+ skip_bias)
 * The accumulators take the 3x3 epilogue, change to the skip operands' scale (a power of two: exact) and go on as the accumulators of
 * the skip convolution over skip_cin more input channels (one tap): no separate 1x1 kernel, no write and re-read of its result.
 * Against the two-kernel form (slr_conv1x1_forward -> residual) the result differs by the order of the last additions only (fp32
 * rounding; tests/test_gpu_parity.py).  skip_wsplit: slr_conv1x1_split_weights(Cout, skip_cin, skip_wscale); the skip input shares
 * `xscale`.  Split-f16 rung only (no SLR_CONV_F32 / _WINO), main input and skip input channel-blocked (SLR_CONV_IN_B8 and
 * SLR_CONV_SKIP_B8 both set, skip_cin % 8 == 0), Cout > 4; anything else is SLR_E_BADARG -- callers keep the two-kernel form there
 * (the networks' 3-channel first blocks and 65- / 3-channel ends).
 * With SLR_CONV_POOL_OUT (and SLR_CONV_OUT_B8, Cout > 64) the "Down" block's nn.AvgPool2d(3, stride=2, padding=1) (blocks.py:196-199)
 * happens in the epilogue: the full-resolution result -- whose only reader is the pool -- is never written; `out` is the pooled tensor,
 * um_out stays full resolution.  A wave pools the 4 x 16 pixels of its tile in registers; the pooled pixels that need the row above /
 * the column left of the tile are completed by a small second launch from side buffers in pool_ws (slr_conv_pool_ws_bytes: the last row
 * of every tile row and the last column of every tile column, ~16 % of the full-resolution tensor).  Same 9 terms per pooled pixel as
 * slr_avgpool3x3s2, summed rows first: equal to the two-kernel form to fp32 rounding.  pool_ws = NULL otherwise. */
size_t slr_conv_pool_ws_bytes(int N, int Cout, int H, int W);
/* With SLR_CONV_UP_OUT (same conditions) the "Up" block's nn.Upsample(scale_factor=2, mode='bilinear') (blocks.py:200-203) happens in the
 * epilogue: a tile's 8 x 32 pixels give the 16 x 64 output pixels below them; all but the first / last row and column of that block come
 * from the wave's registers and neighbouring lanes, those borders are written by a second launch from side buffers (pool_ws of
 * slr_conv_up_ws_bytes: first and last row / column of every tile).  Expression and order of slr_upsample_bilinear2x: bit-identical to
 * the two-kernel form.  The low-resolution result is never written; um_out stays at the convolution's resolution. */
size_t slr_conv_up_ws_bytes(int N, int Cout, int H, int W);
int slr_conv3x3_forward_skip(const float *in, const void *wsplit, const float *bias, float *out,
                             int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                             const float *pre_scale, const float *pre_shift,
                             const float *skip_in, const void *skip_wsplit, const float *skip_bias /* [Cout] or NULL */, int skip_cin,
                             float skip_wscale, void *pool_ws, size_t pool_ws_bytes, int layout, void *stream);
int slr_pconv3x3_forward_skip(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                              const void *wsplit, float wscale, float xscale, const float *bias, float *out, float *um_out,
                              int N, int Cin, int Cout, int H, int W,
                              const float *skip_in, const void *skip_wsplit, int skip_cin, float skip_wscale,
                              void *pool_ws, size_t pool_ws_bytes, int layout, void *stream);

/* Cout <= 4 (the 128 -> 3 end of the decoders), channel-blocked input: the first convolution of the block as slr_conv3x3_forward /
 * slr_pconv3x3_forward, and next to it skip_out [N,Cout,H,W] = conv1x1(in) + skip_bias -- the block's skip branch on the SAME (raw) input
 * (blocks.py:192-193, 243-247), which the caller hands to the block's second convolution as its residual.  The 128 input planes are read
 * from HBM once instead of twice.  skip_w4: plain fp32 weights [Cin][4] (row ci = w[0..3][ci], zero padded), 16-byte aligned.  Either rung
 * (the <= 4-channel kernel is fp32 FMAs on both).  Bit-identical to slr_conv1x1_small on the same input. */
int slr_conv3x3_forward_skipout(const float *in, const void *wsplit, const float *bias, const float *residual, float *out,
                                int N, int Cin, int Cout, int H, int W, float wscale, float xscale,
                                const float *pre_scale, const float *pre_shift,
                                const float *skip_w4, const float *skip_bias /* [Cout] or NULL */, float *skip_out, int layout, void *stream);
int slr_pconv3x3_forward_skipout(const float *x, const float *pre_scale, const float *pre_shift, const float *mask,
                                 const void *wsplit, float wscale, float xscale, const float *bias, const float *residual,
                                 const float *next_scale, const float *next_shift, float *out, float *um_out,
                                 int N, int Cin, int Cout, int H, int W,
                                 const float *skip_w4, float *skip_out, int layout, void *stream);

/* 1x1 convolution (skip branch of the residual blocks, models/layers/blocks.py:192-193,243-247) on the same
 * split-f16 arithmetic: out = conv1x1(in) + bias.  HBM-bound, no LDS.  Weights prepared once per layer with
 * slr_conv1x1_split_weights into slr_conv1x1_weight_bytes(Cout, Cin) bytes; wscale as for the 3x3 kernel. */
size_t slr_conv1x1_weight_bytes(int Cout, int Cin);
int slr_conv1x1_split_weights(const float *w /* [Cout,Cin,1,1] */, void *wsplit, int Cout, int Cin,
                              float wscale, void *stream);
int slr_conv1x1_f32_weights(const float *w /* [Cout,Cin,1,1] */, void *wfrag /* slr_conv1x1_weight_bytes */, int Cout, int Cin,
                            void *stream);
int slr_conv1x1_forward(const float *in, const void *wsplit, const float *bias /* [Cout] or NULL */, float *out,
                        int N, int Cin, int Cout, int H, int W, float wscale, float xscale, int layout, void *stream);

/* ------------------------------------------------------------------ decoder resampling stages (8 f3) */

/* nn.AvgPool2d(3, stride=2, padding=1) (count_include_pad): "Down" of models/layers/blocks.py:196-199.
 *   in [N,C,H,W] -> out [N,C,(H-1)/2+1,(W-1)/2+1] */
int slr_avgpool3x3s2(const float *in, float *out, int N, int C, int H, int W, int b8 /* both tensors channel-blocked */,
                     void *stream);

/* nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False): "Up" of blocks.py:200-203.
 *   in [N,C,H,W] -> out [N,C,2H,2W] */
int slr_upsample_bilinear2x(const float *in, float *out, int N, int C, int H, int W, int b8 /* both tensors channel-blocked */,
                            void *stream);

/* 1x1 convolution onto 1..4 output channels (the skip branch of the decoder's last block,
 * blocks.py:192-193,243-247 with configs.py:117-137): out = bias + w . in;  w [Cout,Cin], bias [Cout] or NULL.
 *   NCHW input requires H*W % 4 == 0 and 16-byte aligned tensors. */
int slr_conv1x1_small(const float *in, const float *w, const float *bias, float *out,
                      int N, int Cin, int Cout, int H, int W, int in_b8 /* `in` channel-blocked; `out` is always NCHW */,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SLR_SPLAT_H */
