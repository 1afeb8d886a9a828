// splat_core.hpp -- device building blocks shared by the tile kernels (gfx950): buffer-descriptor memory access, wave / workgroup
// scans, and the two phases every tile kernel has in common once a piece's entries are known:
//   records   per-OUTPUT-pixel lists of (entry, weight) built in LDS with integer atomics (once per entry, not per channel);
//   stream    the chunk pipeline: stage the entries' source values plane chunk by plane chunk in LDS (loads two chunks ahead),
//             every work-item gathers its own pixel in registers, normalises, stores -- every output byte written once.
// Where the instructions go (rocprofv3 SQ counters on the fused kernel, round 3 build: 61 M VALU wave-instructions per frame = 60 % of
// every SIMD's cycles) decided the shape of this code: addresses live in buffer descriptors + 32-bit offsets (no 64-bit vector
// address arithmetic per load / store), the normaliser is ONE reciprocal per output pixel (an IEEE division is ~11 instructions, and
// there were C of them per pixel), stores outside the image or past the last plane are dropped by the descriptor's range check or a
// scalar branch (no trash tile, no pointer selects), nothing in the chunk loop reads kernel arguments from memory.
#pragma once
#include "slr_common.hpp"
#include "splat_types.hpp"

namespace slr {

// ---- memory access through buffer descriptors ------------------------------------------------------------------------------
// address = base + soffset (SGPR) + voffset (VGPR, 32-bit); a voffset at or past num_records is out of range: loads return 0,
// stores are dropped.  The descriptor must be wave-uniform (built from kernel arguments / block-index arithmetic only).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t BUF_OOB = 0x80000000u;          // a voffset no descriptor of this library covers (num_records < 2^31)

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// 16 bytes at a 16-byte aligned offset.  (The whole vector is bit-cast before its elements are taken: with the elements of the builtin's
// integer vector bit-cast one by one, this compiler narrows the load to ONE dword and splats it -- tests/test_abi_and_host.py counts the
// buffer_load_dwordx4 of the clip kernels.)
typedef unsigned int buf_u32x4 __attribute__((vector_size(16)));
typedef float buf_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, uint32_t voff, uint32_t soff) {
    const buf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const buf_f32x4 f = __builtin_bit_cast(buf_f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ void buf_st4(rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {     // 16 bytes at a 16-byte aligned offset
    const buf_f32x4 f = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(buf_u32x4, f), r, voff, soff, SLR_STORE_AUX);
}
template <int AUX = SLR_STORE_AUX>
__device__ __forceinline__ void buf_st(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, AUX);
}
// agent scope (sc1): a store that is written through to memory / a load that does not take another workgroup's data from this XCD's L2
// -- what workgroups on different XCDs exchange through (their L2s are not coherent with each other)
constexpr int BUF_SC1 = 16;
__device__ __forceinline__ float buf_ld_sc1(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, BUF_SC1));
}

// ---- scans --------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan over the TILE_PIX work-items of a workgroup (one value each); wsum: [TILE_PIX / 64] words of LDS.
// Contains one barrier; the caller synchronises before wsum is reused.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, int tid) {
    const int lane = tid & 63, wid = tid >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < TILE_PIX / 64; ++w) woff += (w < wid) ? wsum[w] : 0u;
    return woff + inc - v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---- normalisation (softsplat.py:684-686 / animating_softmax_splating.py:923-924) ---------------------------------------------
__device__ __forceinline__ float norm_divisor(float nrm, int norm_mode, float eps) {
    return norm_mode == SLR_NORM_ZERO_TO_ONE ? (nrm == 0.0f ? 1.0f : nrm) : fmaxf(nrm, eps);
}

}  // namespace slr
