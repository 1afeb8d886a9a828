// splat_queue.hpp -- work distribution of the PERSISTENT tile kernels (gfx950).
// A launch has as many workgroups as the chip holds (2 or 3 per CU, by the kernel's LDS); each keeps its slot and pulls work items from
// ticket counters until none is left.  Why not one workgroup per item: traced with the constant clock (round 4), a slot stays empty
// ~4.5 us between the end of one workgroup and the first instruction of the next, and a launch's tail is whatever the dispatcher's
// last round leaves; a pulling workgroup starts its next item while its last stores are still in flight.
//   * one head per XCD (a 64-byte line each): workgroup on XCD x takes tickets from head x -- the item order of a queue keeps
//     neighbouring tiles, and the same tile of the consecutive frames of a batch, on ONE XCD's L2 (the placement the per-item launches
//     got from blockIdx % 8); when its own queue is empty it steals from the next XCD's (x + 1, x + 2, ...), so nobody idles while
//     work is left anywhere;
//   * the ticket after the one being worked on is already in flight (one returning agent-scope atomic per item, issued by work-item 0
//     at the start of an item, consumed at its end: a dequeue costs 1.1 - 1.3 us under load, MI355X_MICROARCH.md "dequeue");
//   * the last workgroup to leave resets the heads: the counters live in caller-owned scratch (slr_splat_queue_bytes(), zero before the
//     first launch, left zero by every launch -- HIP-graph replays included).  One launch at a time per scratch buffer.
// Placement is a speed matter only: any workgroup may run any item (HW_REG_XCC_ID merely picks the first queue).
#pragma once
#include "splat_core.hpp"

namespace slr {

constexpr uint32_t Q_XCD = 8;                       // queues (one per XCD)
constexpr uint32_t Q_LINE = 16;                     // words per counter: every head on its own 64-byte line
constexpr size_t QUEUE_BYTES = (size_t)(Q_XCD + 1) * Q_LINE * 4;      // 8 heads + the exit counter

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & (Q_XCD - 1u);
}

struct Puller {
    uint32_t *heads;                                // [Q_XCD + 1][Q_LINE]
    uint32_t per_q;                                 // tickets per queue
    uint32_t xcc, qi;                               // the queue tried first; queues found empty so far
    uint32_t pend;                                  // (work-item 0) the ticket in flight
};

// slot: one LDS word nobody else uses.  Contains barriers: every work-item of the workgroup calls these.
__device__ __forceinline__ void pull_begin(Puller &P, uint32_t *heads, uint32_t per_q, int tid) {
    P.heads = heads; P.per_q = per_q; P.xcc = xcc_id(); P.qi = 0u; P.pend = 0u;
    if (tid == 0) P.pend = atomicAdd(&heads[P.xcc * Q_LINE], 1u);
}

// -> true: `ticket` of queue `queue` is this workgroup's next item (and the ticket after it is in flight); false: every queue is empty.
__device__ __forceinline__ bool pull_next(Puller &P, uint32_t *slot, int tid, uint32_t &ticket, uint32_t &queue) {
#if SLR_NO_TICKET_AHEAD
    if (tid == 0) *slot = atomicAdd(&P.heads[((P.xcc + P.qi) & (Q_XCD - 1u)) * Q_LINE], 1u);
#else
    if (tid == 0) *slot = P.pend;
#endif
    __syncthreads();
    uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
    __syncthreads();                                // (slot may be rewritten below / by the next call)
    while (k >= P.per_q) {                          // this queue is empty: the next XCD's
        if (++P.qi == Q_XCD) return false;
        if (tid == 0) *slot = atomicAdd(&P.heads[((P.xcc + P.qi) & (Q_XCD - 1u)) * Q_LINE], 1u);
        __syncthreads();
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
        __syncthreads();
    }
    queue = (P.xcc + P.qi) & (Q_XCD - 1u);
#if !SLR_NO_TICKET_AHEAD
    if (tid == 0) P.pend = atomicAdd(&P.heads[queue * Q_LINE], 1u);
#endif
    ticket = k;
    return true;
}

// Every workgroup calls this once, after its last pull_next returned false (no atomic of its own is in flight any more).
__device__ __forceinline__ void pull_end(const Puller &P, int tid, uint32_t workgroups) {
    if (tid == 0 && atomicAdd(&P.heads[Q_XCD * Q_LINE], 1u) == workgroups - 1u) {
#pragma unroll
        for (uint32_t q = 0; q <= Q_XCD; ++q) P.heads[q * Q_LINE] = 0u;      // everybody has left: zero for the next launch
    }
}

// ticket of queue q -> item: groups of SLR_XCD_GROUP consecutive items per queue, the groups of a batch's frames interleaved:
//   ticket k: sub = k % G, frame = (k / G) % nb, group = k / (G * nb);  item = (group * 8 + q) * G + sub
// (nb = 1: the order xcd_item() gave the per-item launches).  rcp = ceil(2^32 / nb): exact for k / G < 2^16.
__device__ __forceinline__ void ticket_item(uint32_t k, uint32_t q, uint32_t nb, uint32_t rcp, uint32_t &frame, uint32_t &item) {
    constexpr uint32_t G = SLR_XCD_GROUP;
    const uint32_t sub = k % G, kg = k / G;
    const uint32_t grp = nb > 1u ? (uint32_t)(((unsigned long long)kg * rcp) >> 32) : kg;
    frame = kg - grp * nb;
    item = (grp * Q_XCD + q) * G + sub;
}
// tickets a queue needs to cover `items` items per frame of nb frames
__host__ __device__ __forceinline__ uint32_t tickets_per_queue(uint32_t items, uint32_t nb) {
    constexpr uint32_t G = SLR_XCD_GROUP;
    return ((items + Q_XCD * G - 1u) / (Q_XCD * G)) * G * nb;
}

}  // namespace slr
