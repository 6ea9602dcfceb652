// The tiled form of the LC loss as a device function: shared by its stand-alone launch (lc_loss.hip) and the dense pose unit
// (lc_fused_dense.hip: loss and solve workgroups in one grid).
#pragma once
#include "lc_loss_body.h"

#ifndef LC_GRID_TICKETS
#define LC_GRID_TICKETS 1  // A/B switch (scripts/ubench/tiled_loss.py): 0 = (sample, tile) from blockIdx (relies on in-order dispatch)
#endif

namespace lc {
namespace loss {

// Workgroups take (sample, slice) from a ticket counter as they start (see grid_arrive_wait for why that makes the hand-off
// deadlock-free -- whatever else shares the grid: the tickets handed out are a prefix of the LOSS workgroups in start order); the
// last workgroup of a sample to finish zeroes the sample's counters, the last sample the header, so the workspace is left as it was
// found (all zero) for the next launch on the same stream.  block: this workgroup's index among the loss workgroups (used only
// without tickets).
template <bool COV2D>
__device__ __forceinline__ void tiled_workgroup(const LossParams& p, int T, int S, int TS, LossSharedLoop& sh, unsigned& ticket_sh, unsigned block) {
    unsigned* head = static_cast<unsigned*>(p.workspace);
#if LC_GRID_TICKETS
    if (threadIdx.x == 0) ticket_sh = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = ticket_sh;
#else
    const unsigned ticket = block;
#endif
    const int b = (int)(ticket / (unsigned)S);
    GridCtx g;
    g.head = head;
    g.ctr = head + 4 + 2 * (size_t)b;
    g.rows = reinterpret_cast<double*>(static_cast<char*>(p.workspace) + grid_rows_offset_bytes(p.B)) + (size_t)b * T * kGridRow;
    g.T = T;
    g.S = S;
    g.TS = TS;
    g.slice = (int)(ticket % (unsigned)S);
    g.timed_out = 0;
    sample<false, COV2D, true, LossSharedLoop>(p, b, sh, &g);
    // retire: relaxed device-scope atomics only (no cache maintenance) -- a workgroup counts itself finished after the hand-off,
    // the counters it may then zero are touched by nobody else any more
    if (threadIdx.x == 0) {
        // A hand-off that timed out (it cannot: tickets) poisons the sample whichever slice it happened in: the slice counts itself out
        // with a flag in the counter's upper half, and the LAST workgroup of the sample to retire -- every slice, slice 0 and its finite
        // loss included, is past its stores by then -- overwrites loss[b] with NaN (written through, like the counters).  The slice's own
        // gradient rows were formed from incomplete sums; the NaN loss is what tells the caller (never silently wrong).  head[2] counts the
        // time-outs of the workspace's lifetime and is NOT reset: a caller that sees it non-zero (or a NaN loss) re-zeroes the workspace.
        const unsigned d = __hip_atomic_fetch_add(g.ctr + 1, g.timed_out ? 0x10001u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((d & 0xFFFFu) == (unsigned)S - 1u) {  // every workgroup of the sample is past the hand-off
            if ((d >> 16) || g.timed_out) __hip_atomic_store(p.loss + b, __builtin_nanf(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned s = __hip_atomic_fetch_add(head + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == (unsigned)p.B - 1u) {  // every workgroup of the grid has taken its ticket
                __hip_atomic_store(head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(head + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace loss
}  // namespace lc
