// Latency of a one-word hand-off between two workgroups through global memory on MI355X: a ping-pong of N round trips between workgroup 0 and
// workgroup `peer` (peer = 8: same XCD under the round-robin dispatch, peer = 1: the neighbouring XCD), with the accesses at agent scope
// (sc1: what lc_common.h's xcd_load / xcd_store use) or at workgroup scope (sc0: served by the XCD's own L2 -- only meaningful on one XCD).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/pingpong.cpp -o /tmp/pingpong && /tmp/pingpong
#include <hip/hip_runtime.h>

#include <cstdio>

template <int SCOPE>
__device__ unsigned ld(const unsigned* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ void st(unsigned* q, unsigned v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, SCOPE); }

template <int SCOPE>
__global__ void pingpong(unsigned* words, int peer, int rounds, unsigned long long* ticks, unsigned* xcc) {
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)peer ? 1 : -1);
    if (me < 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me] = id;
    unsigned* mine = words + 64 * me;
    unsigned* other = words + 64 * (1 - me);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    long spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            st<SCOPE>(mine, (unsigned)r);
            while (ld<SCOPE>(other) != (unsigned)r)
                if (++spins > (1l << 26)) { ticks[me] = ~0ull; return; }
        } else {
            while (ld<SCOPE>(other) != (unsigned)r)
                if (++spins > (1l << 26)) { ticks[me] = ~0ull; return; }
            st<SCOPE>(mine, (unsigned)r);
        }
    }
    ticks[me] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
    unsigned *words, *xcc;
    unsigned long long* ticks;
    hipMalloc(&words, 1024);
    hipMalloc(&xcc, 8);
    hipMalloc(&ticks, 16);
    const int rounds = 2000;
    for (int peer : {8, 1, 16, 9}) {
        for (int scope = 0; scope < 2; ++scope) {
            hipMemset(words, 0, 1024);
            hipMemset(ticks, 0, 16);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (scope == 0) pingpong<__HIP_MEMORY_SCOPE_AGENT><<<32, 64>>>(words, peer, rounds, ticks, xcc);
            else pingpong<__HIP_MEMORY_SCOPE_WORKGROUP><<<32, 64>>>(words, peer, rounds, ticks, xcc);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t[2];
            unsigned x[2];
            hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
            hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
            std::printf("workgroups 0 and %2d (XCC_ID %u and %u), %s scope: %s, %.0f ns per round trip (two hand-offs)\n", peer, x[0] & 15, x[1] & 15,
                        scope == 0 ? "agent    " : "workgroup", t[0] == ~0ull || t[1] == ~0ull ? "NEVER ARRIVED" : "ok", ms * 1e6 / rounds);
        }
    }
    return 0;
}
