// conv_few.hpp -- 3x3 convolution onto AT MOST 4 output channels (the 128 -> 3 end of the decoders, blocks.py:173-248 /
// architectures.py:345-375; the 3 -> 3 and n -> 1 heads of the small networks): same arguments, prologue, partial-convolution epilogue,
// mask update and residual as conv3x3_split_kernel (conv.hip), but the products are plain fp32 FMAs on the vector ALUs.
//
// Why not the matrix cores: the narrowest MFMA tile of conv3x3_split_kernel is 32 output channels -- 3 useful ones cost 32 (x 3 MFMAs per
// product on the split rung): 259 us per 768x1280 frame on the split rung, 571 us on the fp32 rung.  3 channels x 9 taps x 128 inputs are
// 3456 FMAs per output pixel = 6.8 GFLOP per frame, 86 us of the chip's unpacked fp32 vector rate, with the reference's own arithmetic
// (fp32 operands, fp32 products; no operand scales, no magnitude limit, nothing to saturate) -- so BOTH rungs take this kernel.
//
// Workgroup = 256 work-items = a 16 x 64 block of output pixels; a work-item owns 4 consecutive pixels of a row x all output channels
// (12 accumulators).  Per chunk of 8 input channels the (16+2) x (64+2) halo block is staged through registers (loads of chunk c + 1 in
// flight under the FMAs of chunk c) into ONE LDS buffer as planar floats (row stride 68: a row of a work-item's 6 values is one 16-byte
// and one 8-byte read), the chunk's 72 x 4 weights sit in LDS as well (plain [ci][tap][4] floats: slr_conv3x3_*_weights
// write this layout when Cout <= 4; broadcast 16-byte reads -- through scalar loads the same weights cost a full wait per group of
// SGPRs: 245 us), and a channel is 18 LDS values + 9 weight reads for 108 FMAs.
#pragma once

namespace slr {

constexpr int CF_MAXCO = 4;                              // output channels this kernel covers
constexpr int CF_BW = 64, CF_BH = 16, CF_THREADS = 256;  // output block, work-items
constexpr int CF_HW = CF_BW + 2, CF_HH = CF_BH + 2;      // halo block
constexpr int CF_NPX = CF_HW * CF_HH;                    // 1188 halo pixels
constexpr int CF_STR = 68;                               // floats per staged row (16-byte aligned rows)
constexpr int CF_ITEMS = (CF_NPX + CF_THREADS - 1) / CF_THREADS;   // staging items (halo pixel x 8 channels) per work-item and chunk

__host__ __device__ inline int conv_few_cin_pad(int Cin) { return (Cin + 7) / 8 * 8; }

// w [Cout,Cin,3,3] fp32 -> [ci padded to 8][tap][4] fp32, zeros for the padding
__global__ __launch_bounds__(256) void conv_few_weights_kernel(const float *__restrict__ w, float *__restrict__ wf, int Cout, int Cin, int CinP) {
    const int total = CinP * 36;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i & 3, tap = (i >> 2) % 9, ci = i / 36;
        wf[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
    }
}

// NCO: output channels computed (>= Cout)
// SKP: the block's 1x1 skip convolution of the SAME input (blocks.py:192-193, 243-247: conv_b(x) next to conv_aa(relu(bn(x)))) is computed
// alongside and written to a.skip_out [N,Cout,H,W]: the input is read once instead of twice (the skip kernel's 128 planes were 1.6 % of a
// frame).  The staged block holds the ACTIVATED input; the raw values pass through the registers of the STAGING work-item, so that one sums
// the skip of its (up to 5) halo pixels, chunk by chunk, and writes those that lie inside the block at the end: no second read, no LDS.
// (First form: every work-item read the raw values of its own 4 output pixels again, L2 hits, one chunk ahead -- the kernel is bound by its
// loads and took exactly the time of the skip kernel it replaced longer: 635 -> 970 us per call of 4 frames.)  Bias first, then fused
// multiply-adds in ascending channel order -- bit-identical to slr_conv1x1_small.
template <int NCO, bool PRE, bool INB8, bool SKP = false>
__global__ __launch_bounds__(CF_THREADS, SKP && NCO < 4 ? 3 : 2) void conv3x3_few_kernel(ConvArgs a) {      // (the plain kernel takes 121 registers: 3 workgroups per CU by its 53 KB of LDS; SKP must stay within 170 for the same -- 4 output channels do not)
    static_assert(!SKP || INB8, "the skip output reads a channel-blocked input");
    __shared__ __attribute__((aligned(16))) float xs[8][CF_HH][CF_STR];
    __shared__ float4 wl[72 + (SKP ? 16 : 0)];           // the chunk's weights: [8 ci][9 taps] x 4 output channels (+ SKP: [2 buffers][8 ci] x 4 of the skip)
    __shared__ float mpl[CF_NPX];                        // mask plane over the halo block (0 outside the image)
    __shared__ __attribute__((aligned(16))) float pss[PRE ? 2 * CV_MAXCIN : 4];       // prologue scale / shift per input channel
    const int tid = threadIdx.x;
    const int tiles_x = (a.W + CF_BW - 1) / CF_BW;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int x0 = tx * CF_BW, y0 = ty * CF_BH;
    const int n = blockIdx.z;
    const int HW = a.H * a.W;
    const int cmax = a.Cin - 1;
    const int nch = conv_few_cin_pad(a.Cin) >> 3;
    const float *inb = a.in + (size_t)n * a.Cin * HW;
    const float *wf = reinterpret_cast<const float *>(a.w);
    const int pre = a.pre;
    const bool nonzero_mask = !SKP && pre == PRE_BN_NONZERO;       // (SKP: explicit masks only -- the entry points check; frees the count registers)

    if (PRE) {
        for (int i = tid; i < nch * 8; i += CF_THREADS) {           // padded channels: scale = shift = 0 -> 0
            pss[i] = i < a.Cin ? a.pre_scale[i] : 0.0f;
            pss[CV_MAXCIN + i] = i < a.Cin ? a.pre_shift[i] : 0.0f;
        }
    }
    // staging items of this work-item: halo pixels tid, tid + 256, ... (the same pixels in every chunk)
    int off[CF_ITEMS];
    float mk0[CF_ITEMS], cnt[CF_ITEMS];
    int lds_at[CF_ITEMS];
#pragma unroll
    for (int i = 0; i < CF_ITEMS; ++i) {
        const int p = tid + i * CF_THREADS;
        const int pr = p / CF_HW, pc = p - pr * CF_HW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        const bool live = p < CF_NPX;
        const bool ok = live & (gy >= 0) & (gy < a.H) & (gx >= 0) & (gx < a.W);
        off[i] = ok ? gy * a.W + gx : 0;
        const float mv = (a.mask && ok) ? a.mask[(size_t)n * HW + off[i]] : 0.0f;
        if (a.mask && live) mpl[p] = mv;                   // for the 3x3 box sum of the epilogue
        mk0[i] = ok ? (pre == PRE_BN_MASK ? mv : 1.0f) : 0.0f;
        cnt[i] = 0.0f;
        lds_at[i] = live ? pr * CF_STR + pc : -1;
    }
    const int c8max = (a.Cin >> 3) - 1;                   // INB8: last 8-channel group
    float st[CF_ITEMS][8];
    float4 wreg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int row = tid >> 4, xq = (tid & 15) * 4;        // this work-item's output pixels: (y0 + row, x0 + xq .. + 3)
    float sacc[CF_ITEMS][NCO];                           // SKP: skip sums of this work-item's staging pixels
    if (SKP) {
#pragma unroll
        for (int i = 0; i < CF_ITEMS; ++i)
#pragma unroll
            for (int co = 0; co < NCO; ++co) sacc[i][co] = (a.skip_bias && co < a.Cout) ? a.skip_bias[co] : 0.0f;
    }
    auto load_chunk = [&](int c) {
        if (tid < 72) wreg = reinterpret_cast<const float4 *>(wf)[c * 72 + tid];
        if (SKP && tid >= 72 && tid < 80) wreg = reinterpret_cast<const float4 *>(a.skip_w)[min(c + 1, nch - 1) * 8 + tid - 72];   // (one chunk ahead: read before the barrier)
#pragma unroll
        for (int i = 0; i < CF_ITEMS; ++i) {
            if (INB8) {
                const float4 *q = reinterpret_cast<const float4 *>(inb) + ((size_t)min(c, c8max) * HW + (unsigned)off[i]) * 2;
                const float4 u = q[0], v = q[1];
                st[i][0] = u.x; st[i][1] = u.y; st[i][2] = u.z; st[i][3] = u.w;
                st[i][4] = v.x; st[i][5] = v.y; st[i][6] = v.z; st[i][7] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) st[i][j] = inb[(size_t)min(c * 8 + j, cmax) * HW + (unsigned)off[i]];   // (past Cin: zero weights)
            }
        }
    };
    // prologue (normalization.py:231, ReLU, partialconv2d.py:69) + LDS store; the same expression as conv3x3_split_kernel's stage_value
    auto store_chunk = [&](int c) {
        if (tid < 72) wl[tid] = wreg;
        if (SKP) {
            // the skip's 8 channels of this chunk on the raw values in the staging registers; its weights were written a chunk ago
            if (tid >= 72 && tid < 80) wl[72 + ((c + 1) & 1) * 8 + tid - 72] = wreg;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w4 = wl[72 + (c & 1) * 8 + j];
                const float wt[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int i = 0; i < CF_ITEMS; ++i)
#pragma unroll
                    for (int co = 0; co < NCO; ++co) sacc[i][co] = __builtin_fmaf(wt[co], st[i][j], sacc[i][co]);
            }
        }
        float sc[8], sh[8];                                // the chunk's scale / shift: four 16-byte broadcast reads, not two reads per value
        if (PRE) {
            const float4 s0 = *reinterpret_cast<const float4 *>(&pss[c * 8]), s1 = *reinterpret_cast<const float4 *>(&pss[c * 8 + 4]);
            const float4 h0 = *reinterpret_cast<const float4 *>(&pss[CV_MAXCIN + c * 8]), h1 = *reinterpret_cast<const float4 *>(&pss[CV_MAXCIN + c * 8 + 4]);
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        }
#pragma unroll
        for (int i = 0; i < CF_ITEMS; ++i) {
            if (lds_at[i] < 0) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = st[i][j];
                float v;
                if (PRE) {
                    const int ch = c * 8 + j;
                    const float mk = (nonzero_mask & (x == 0.0f)) ? 0.0f : mk0[i];
                    v = fmaxf(x * sc[j] - sh[j], 0.0f) * mk;
                    cnt[i] += (ch <= cmax) ? mk : 0.0f;    // (x != 0) inside the image, real channels only: the derived mask's channel sum
                } else {
                    v = x * mk0[i];
                }
                (&xs[j][0][0])[lds_at[i]] = v;
            }
        }
    };

    float acc[NCO][4];
#pragma unroll
    for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = 0.0f;

    if (SKP && tid >= 72 && tid < 80) wl[72 + tid - 72] = reinterpret_cast<const float4 *>(a.skip_w)[tid - 72];      // chunk 0's (visible behind the first barrier)
    load_chunk(0);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();                                   // the previous chunk has been read (first round: pss / mpl written)
        store_chunk(c);
        __syncthreads();
        if (c + 1 < nch) load_chunk(c + 1);                // in flight under this chunk's FMAs

#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float4 xa = *reinterpret_cast<const float4 *>(&xs[j][row + dy][xq]);
                const float2 xb = *reinterpret_cast<const float2 *>(&xs[j][row + dy][xq + 4]);
                const float x[6] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float4 w4 = wl[j * 9 + dy * 3 + dx];     // (broadcast read: the chunk's 72 taps x 4 channels)
                    const float wt[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int co = 0; co < NCO; ++co) {
                        const float wv = wt[co];
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[co][p] = __builtin_fmaf(x[p + dx], wv, acc[co][p]);
                    }
                }
            }
        }
    }
    if (nonzero_mask) {                                    // mask plane = channel sum of (x != 0) (architectures.py:369, partialconv2d.py:61)
#pragma unroll
        for (int i = 0; i < CF_ITEMS; ++i)
            if (lds_at[i] >= 0) mpl[tid + i * CF_THREADS] = cnt[i];
    }
    __syncthreads();

    // epilogue: the operations of conv3x3_split_kernel's, in its order
    const bool partial = a.partial != 0, has_bias = a.bias != nullptr, has_res = a.residual != nullptr, has_next = a.next_scale != nullptr;
    const int oy = y0 + row, ox0 = x0 + xq;
    const bool rowok = oy < a.H;
    const bool vec = rowok && (a.W % 4 == 0) && (ox0 + 3 < a.W) && !(((uintptr_t)a.out | (uintptr_t)a.residual | (uintptr_t)a.um_out) & 15);
    float um[4], ratio[4];
    if (partial) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float box = 0.0f;                              // conv(mask, ones): 3x3 box sum, zero padded (:61)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) box += mpl[(row + dy) * CF_HW + xq + p + dx];
            const float u = box * a.mask_scale;            // exact small integers in fp32
            um[p] = fminf(fmaxf(u, 0.0f), 1.0f);
            ratio[p] = (1.0f / (u + 1e-8f)) * a.winsize * um[p];       // torch: scalar / tensor = reciprocal * scalar
        }
        if (a.um_out) {
            float *up = a.um_out + (size_t)n * HW + (rowok ? (size_t)oy * a.W : 0) + ox0;
            if (vec) *reinterpret_cast<float4 *>(up) = make_float4(um[0], um[1], um[2], um[3]);
            else
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (rowok && ox0 + p < a.W) up[p] = um[p];
        }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
        if (co >= a.Cout) break;                           // (uniform)
        float o[4];
        const float eb = has_bias ? a.bias[co] : 0.0f;
        const size_t base = ((size_t)n * a.Cout + co) * HW + (rowok ? (size_t)oy * a.W : 0) + ox0;
        float rv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (has_res) {
            if (vec) { const float4 r4 = *reinterpret_cast<const float4 *>(a.residual + base); rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w; }
            else
#pragma unroll
                for (int p = 0; p < 4; ++p) rv[p] = (rowok && ox0 + p < a.W) ? a.residual[base + p] : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            o[p] = acc[co][p];
            if (partial) {
                o[p] = (o[p] * ratio[p] + eb) * um[p];                                              // :72-74
                if (has_res) o[p] += rv[p];                                                         // blocks.py:248
                if (has_next) o[p] = fmaxf(o[p] * a.next_scale[co] - a.next_shift[co], 0.0f) * um[p];   // blocks.py:233-236
            } else {
                if (has_bias) o[p] += eb;
                if (has_res) o[p] += rv[p];
            }
        }
        if (vec) *reinterpret_cast<float4 *>(a.out + base) = make_float4(o[0], o[1], o[2], o[3]);
        else
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (rowok && ox0 + p < a.W) a.out[base + p] = o[p];
    }
    if (SKP) {                                             // the staging pixels inside the block (not its halo) and inside the image
#pragma unroll
        for (int i = 0; i < CF_ITEMS; ++i) {
            const int p = tid + i * CF_THREADS;
            const int pr = p / CF_HW, pc = p - pr * CF_HW;
            const bool inside = p < CF_NPX && pr >= 1 && pr <= CF_BH && pc >= 1 && pc <= CF_BW && y0 - 1 + pr < a.H && x0 - 1 + pc < a.W;
            if (!inside) continue;
#pragma unroll
            for (int co = 0; co < NCO; ++co)
                if (co < a.Cout) a.skip_out[((size_t)n * a.Cout + co) * HW + off[i]] = sacc[i][co];
        }
    }
}

}  // namespace slr
