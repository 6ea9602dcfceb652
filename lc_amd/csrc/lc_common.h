// Shared device helpers for the lc_amd HIP kernels (gfx950 / CDNA4, wave64).
//
// Cross-lane reductions here never touch LDS: the 32- and 16-lane exchanges use gfx950's v_permlane32_swap /
// v_permlane16_swap (a swap of register halves IS the reduce-scatter exchange, so no select is needed), the 8-, 4-, 2-
// and 1-lane exchanges use DPP row_mirror / row_half_mirror / quad_perm moves.
#pragma once
#include <hip/hip_runtime.h>

namespace lc {

// 16-byte store of a write-once stream (a gradient map the kernel never re-reads): non-temporal, so that it does not push the
// maps the next kernel re-reads out of L2 / Infinity Cache (measured on the keypoint head: profiles/r02/head_policy.txt).
#ifndef LC_NT_GRAD_STORES
#define LC_NT_GRAD_STORES 1
#endif
typedef float lc_v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream4(float* q, float a, float b, float c, float d) {
#if LC_NT_GRAD_STORES
    lc_v4f_t r = {a, b, c, d};
    __builtin_nontemporal_store(r, reinterpret_cast<lc_v4f_t*>(q));
#else
    *reinterpret_cast<float4*>(q) = make_float4(a, b, c, d);
#endif
}


constexpr int kWave = 64;

// DPP controls (cdna4 ISA 'DPP_CTRL')
constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppIdentity = 0xE4;      // quad_perm:[0,1,2,3]
constexpr int kDppRowMirror = 0x140;    // lane i <-> 15-i inside each row of 16
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7-i inside each half row of 8

template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

// value of `v` in lane `src` (ds_bpermute: LDS crossbar, no LDS memory)
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const int lo = __shfl(__double2loint(v), src, kWave), hi = __shfl(__double2hiint(v), src, kWave);
    return __hiloint2double(hi, lo);
}

// a' (returned in a) and b' after swapping a's upper 32 lanes with b's lower 32 lanes; a'+b' then holds, in the lower
// half-wave, the two-half sum of a and, in the upper half-wave, the two-half sum of b.
__device__ __forceinline__ double swap32_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
// same for rows of 16: even rows end with the (row, row+1) sum of a, odd rows with that of b
__device__ __forceinline__ double swap16_add(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// lanes with bit 3 set ("up", banks 2-3 of each row) keep hi, the others keep lo; the partner (row mirror) supplies
// the same quantity.  BANK_UP = bank mask of the up lanes: 0xC for the 8-exchange, 0xA for the 4-exchange.
template <int CTRL, int BANK_UP>
__device__ __forceinline__ double dpp_exchange_add(double lo, double hi) {
    constexpr int BANK_DOWN = 0xF & ~BANK_UP;
    const double keep = dpp_mov_f64<kDppIdentity, BANK_UP>(lo, hi);              // up lanes <- hi
    double recv = dpp_mov_f64<CTRL, BANK_UP>(lo, hi);                            // up lanes <- partner's hi
    recv = dpp_mov_f64<CTRL, BANK_DOWN>(recv, lo);                               // down lanes <- partner's lo
    return keep + recv;
}

// All-reduce of K doubles across the 64 lanes of a wave (every lane gets the sum), LDS-free.
template <int K>
__device__ __forceinline__ void wave_allreduce(double (&v)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        double x = v[i];
        x += dpp_mov_f64<kDppQuadXor1>(x, x);
        x += dpp_mov_f64<kDppQuadXor2>(x, x);
        x += dpp_mov_f64<kDppHalfMirror>(x, x);
        x += dpp_mov_f64<kDppRowMirror>(x, x);
        x = swap16_add(x, x);
        x = swap32_add(x, x);
        v[i] = x;
    }
}

// Reduce-scatter of K = 16*R doubles across a wave: after the call lane l holds, in v[0..R-1], the full
// wave sums of entries base..base+R-1 with base = R*(8*b5 + 4*b4 + 2*b3 + b2) (b_i = bit i of l).
// K/2 + K/4 + K/8 + K/16 + 2R exchanges instead of 6K for a butterfly all-reduce, none of them through LDS.
template <int K>
__device__ __forceinline__ void wave_reduce_scatter16(double (&v)[K], int /*lane*/) {
    static_assert(K % 16 == 0, "K must be a multiple of 16");
    constexpr int R = K / 16;
#pragma unroll
    for (int i = 0; i < K / 2; ++i) v[i] = swap32_add(v[i], v[i + K / 2]);
#pragma unroll
    for (int i = 0; i < K / 4; ++i) v[i] = swap16_add(v[i], v[i + K / 4]);
#pragma unroll
    for (int i = 0; i < K / 8; ++i) v[i] = dpp_exchange_add<kDppRowMirror, 0xC>(v[i], v[i + K / 8]);
#pragma unroll
    for (int i = 0; i < K / 16; ++i) v[i] = dpp_exchange_add<kDppHalfMirror, 0xA>(v[i], v[i + K / 16]);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        v[i] += dpp_mov_f64<kDppQuadXor2>(v[i], v[i]);
        v[i] += dpp_mov_f64<kDppQuadXor1>(v[i], v[i]);
    }
}

__device__ __forceinline__ int scatter16_base(int lane, int R) {
    return R * (((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1));
}

// 1/x without the IEEE division sequence: v_rcp_f64 (measured 4.5e-8 relative on gfx950, scripts/ubench/rcp_accuracy.cpp)
// + one Newton step -> 2.2e-15.  x finite, non-zero, normal -- every use is a pivot, a depth or a norm that is checked
// separately.  ~29 cycles for a lone wave against ~72 for `1.0 / x` (scripts/ubench/fp64_latency.cpp).
__device__ __forceinline__ double fast_rcp(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// sqrt(x) and 1/sqrt(x) from v_rsq_f64 + one coupled Goldschmidt step (4e-15 relative; ~39 cycles against ~104 for sqrt()).
// x >= 0; x == 0 gives sqrt 0 / rsqrt +inf.
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& rs) {
    const double r0 = __builtin_amdgcn_rsq(x);
    const double g = x * r0, h = 0.5 * r0;
    const double rr = __builtin_fma(-h, g, 0.5);
    const double g1 = __builtin_fma(g, rr, g), h1 = __builtin_fma(h, rr, h);
    s = x == 0.0 ? 0.0 : g1;
    rs = x == 0.0 ? r0 : 2.0 * h1;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    double s, rs;
    fast_sqrt_rsqrt(x, s, rs);
    return s;
}

// Sum of K doubles per thread over the 64*NW threads of a workgroup, result in every thread, through LDS.
// Thread t stores v[k] at lds[k*LD + t + 2*(t/32)] (32-double segments padded by 16 bytes, LD = 68*NW doubles: conflict-free
// ds_write_b64 and ds_read_b128); thread r < K*S (S = 2*NW segments per row) sums one segment with eight + eight 16-byte
// reads, the S partial sums of a row meet by DPP (S consecutive, S-aligned lanes), the K totals are re-read by all threads.
// Measured cheaper than the permlane/DPP reduce-scatter + broadcast for a lone wave (each 64-bit cross-lane exchange costs
// 25-32 cycles, scripts/ubench/fp64_latency.cpp).  Needs K <= 32, NW in {1, 4} and K*68*NW + K doubles of LDS.
template <int NW>
constexpr int sum_bcast_lds_doubles(int K) { return K * 68 * NW + K; }

// The two halves of the sum can be called separately: block_sum_open() (barrier: the previous totals have been read), then the
// caller stores its K values itself with block_sum_put() AS IT PRODUCES THEM -- the LDS stores of a lone wave drain at about
// 40 B/clk (MI355X_MICROARCH.md, LDS: one wave gets half the store rate), 360 cycles for 28 doubles x 64 lanes, and issued early they
// drain behind the arithmetic that produces the later values -- then block_sum_close() for the reduction and the broadcast.
// Ordering point of the LDS hand-offs inside the block sum.  A workgroup of ONE wavefront needs no barrier and no drain: the
// DS operations of a wave execute in program order (a ds_read issued after a ds_write of another lane's slot returns the new
// value), so the reads simply queue behind the stores and the wave stalls only where it consumes a loaded value; a
// __syncthreads() here costs an `s_waitcnt lgkmcnt(0)` -- a full drain of the LDS queue -- three times per sum.
#ifndef LC_WAVE_SYNC
#define LC_WAVE_SYNC 0  // A/B switch (scripts/ubench/pnp_ab.py): 1 restores the drains for one-wave workgroups
#endif
// Ordering point between the lanes of ONE wavefront that hand data to each other through LDS.  The wave barrier alone is a
// scheduling barrier (IntrNoMem); the wavefront-scope release / acquire pair around it makes the hand-off part of the memory
// model -- the compiler may not move a lane's ds_read above another lane's ds_write across it -- and emits no instruction on
// gfx950 (a wave's DS operations execute in order; checked: identical instruction counts in the ISA with and without the pair).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Values handed from one workgroup to another INSIDE a launch (the workgroups may sit on different XCDs, whose L2s are not
// coherent with each other): written through and read around the caches as agent-scope relaxed atomics (sc1 accesses).  The
// writer orders them before its arrival count with xcd_stores_done() + a workgroup barrier; no cache-wide fence anywhere
// (profiles/r03/NOTES.md 4.1 "Tiled form" has the measurements behind this choice).
template <class T>
__device__ __forceinline__ void xcd_store(T* q, T v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T xcd_load(const T* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xcd_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// "Which workgroup finishes last" for grids of hundreds of workgroups that all finish together: arrivals on ONE word are served one
// after the other (11-13 ns each: 512 of them are 6 us behind a 10 us kernel), so the workgroups count themselves in on one of
// kArrivalShards words, each on a 128-byte line of its own (neighbours in dispatch order -> different words), and only the last of a
// shard goes on to the top word.  `words`: kArrivalWords zeroed unsigned, left zeroed by arrival_reset() (called by the workgroup
// that got `true`, after its last use of what the others wrote).  Called by ONE thread, after xcd_stores_done().
constexpr int kArrivalShards = 16, kArrivalStride = 32;
constexpr int kArrivalWords = (1 + kArrivalShards) * kArrivalStride;
__device__ __forceinline__ bool arrive_is_last(unsigned* words, unsigned block, unsigned blocks) {
    const unsigned shards = blocks < (unsigned)kArrivalShards ? blocks : (unsigned)kArrivalShards;
    const unsigned g = block % shards, members = (blocks - g + shards - 1) / shards;
    if (__hip_atomic_fetch_add(words + (1 + g) * kArrivalStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
    return __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1;
}
__device__ __forceinline__ void arrival_reset(unsigned* words, int tid) {
    if (tid <= kArrivalShards) xcd_store(words + tid * kArrivalStride, 0u);
}

template <int NW>
__device__ __forceinline__ void block_sum_sync() {
    if constexpr (NW == 1 && !LC_WAVE_SYNC) wave_sync();
    else __syncthreads();
}

template <int NW>
__device__ __forceinline__ int block_sum_open(int tid) {
    block_sum_sync<NW>();  // the previous totals have been read
    return tid + 2 * (tid >> 5);
}
template <int NW>
__device__ __forceinline__ void block_sum_put(double* lds, int pos, int k, double v) { lds[k * (68 * NW) + pos] = v; }

template <int K, int NW>
__device__ __forceinline__ void block_sum_close(double (&v)[K], double* lds, int tid);

template <int K, int NW>
__device__ __forceinline__ void block_sum_bcast_lds(double (&v)[K], double* lds, int tid) {
    const int pos = block_sum_open<NW>(tid);
#pragma unroll
    for (int k = 0; k < K; ++k) block_sum_put<NW>(lds, pos, k, v[k]);
    block_sum_close<K, NW>(v, lds, tid);
}

// the reduction half alone: afterwards the K totals sit in LDS at block_sum_totals<K, NW>(lds)[0..K-1], where every thread may
// read them until the next block_sum_open (callers that do not want all K of them in registers at once)
template <int K, int NW>
__device__ __forceinline__ double* block_sum_totals(double* lds) { return lds + K * (68 * NW); }

template <int K, int NW>
__device__ __forceinline__ void block_sum_reduce(double* lds, int tid) {
    static_assert(K <= 32 && (NW == 1 || NW == 4), "unsupported shape");
    constexpr int LD = 68 * NW, S = 2 * NW;
    block_sum_sync<NW>();
    double s = 0;
    if (tid < K * S) {
        const double2* row = reinterpret_cast<const double2*>(lds + (tid / S) * LD + 34 * (tid % S));
        double2 a[8];  // two batches of eight 16-byte reads: keeps the register peak (and so the occupancy) down
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = row[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 t = row[8 + i];
            a[i].x += t.x;
            a[i].y += t.y;
        }
#pragma unroll
        for (int st = 4; st >= 1; st >>= 1) {
#pragma unroll
            for (int i = 0; i < st; ++i) { a[i].x += a[i + st].x; a[i].y += a[i + st].y; }
        }
        s = a[0].x + a[0].y;
    }
    s += dpp_mov_f64<kDppQuadXor1>(s, s);
    if constexpr (NW == 4) {
        s += dpp_mov_f64<kDppQuadXor2>(s, s);
        s += dpp_mov_f64<kDppHalfMirror>(s, s);
    }
    double* tot = lds + K * LD;
    if (tid < K * S && (tid % S) == 0) tot[tid / S] = s;
    block_sum_sync<NW>();
}

template <int K, int NW>
__device__ __forceinline__ void block_sum_close(double (&v)[K], double* lds, int tid) {
    block_sum_reduce<K, NW>(lds, tid);
    const double* tot = block_sum_totals<K, NW>(lds);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
}

// The same sum for FOUR wavefronts without sending every thread's K values through LDS (4 x 64 x K doubles each way: the LDS form
// above is bandwidth-bound there, ~2.0 k cycles per sum of 28): each wavefront reduce-scatters its K values in registers
// (permlane / DPP exchanges, the data halves at every step), 16 of its lanes put its 32 wave totals into LDS, 32 threads add the four
// rows in wave order, every thread reads the totals.  Two barriers per sum: the totals alternate between two LDS rows (`phase`,
// toggled by the call), so a wave still reading the previous totals is never overwritten.  lds: at least 4 * 32 + 2 * 32 doubles.
#ifndef LC_WIDE_SUM_REGS
#define LC_WIDE_SUM_REGS 1  // A/B switch (scripts/ubench/wide_stamps.py): 0 = every thread's values through LDS
#endif
template <int K>
__device__ __forceinline__ void block_sum_waves4(double (&v)[K], double* lds, int tid, int& phase) {
    static_assert(K <= 32, "one reduce-scatter of 32");
    const int lane = tid & 63, wave = tid >> 6;
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < K ? v[k] : 0.0;
    wave_reduce_scatter16<32>(w, lane);
    double* part = lds;                      // [4][32]
    double* tot = lds + 128 + 32 * phase;    // [2][32]
    phase ^= 1;
    if ((lane & 3) == 0) *reinterpret_cast<double2*>(part + 32 * wave + scatter16_base(lane, 2)) = make_double2(w[0], w[1]);
    __syncthreads();
    if (tid < 32) tot[tid] = ((part[tid] + part[32 + tid]) + part[64 + tid]) + part[96 + tid];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
}

// The same sum across the G workgroups that share ONE pose (lc_pnp.hip: lc_pnp_lm_split_kernel -- a batch of few poses with thousands of
// correspondences each leaves three quarters of the chip idle at one workgroup per pose).  Every workgroup reduces its own share as above
// and hands its 32 partial totals to the others through the pose's exchange rows in global memory; each then adds the G rows IN PART ORDER,
// so all of them continue with the same bits and take the same branches of the solve.
// The hand-off is one write and one (polled) read, no counter and no fence: a total travels as two 8-byte words {32 bits of the double,
// 32-bit ticket}, each stored and loaded as ONE agent-scope atomic, and a reader takes a word when it carries the ticket of the sum it is
// waiting for.  Tickets count the sums of a pose over the life of the workspace: `epoch` (a word of the pose's region) holds the ticket of
// the last sum of the previous launch; every part reads it on entry, part 0 writes the new value when it leaves (no part can still be
// waiting to read the old one: part 0 only gets through its first sum after every part has contributed, i.e. has read it).  A slot is
// rewritten every second sum (two rows, by sum parity: a workgroup can be one sum ahead of the slowest, not two -- it cannot finish sum s+1
// before every part has written its s+1 words, i.e. finished reading the rows of sum s), so what a reader finds there is older than the
// ticket it waits for, or it.  Zeroed workspace = ticket 0 everywhere, first sum = ticket 1.
// The wait is bounded in TIME (kSplitWaitTicks of the constant 100 MHz s_memrealtime clock = LC_SPLIT_WAIT_US microseconds, the same budget
// for every split form whatever its polling loop costs; the clock is read once every 16 unsuccessful polls, never on the path of a hand-off
// that arrives) and NOTHING depends on the parts being resident together: a part that has waited
// that long stops for good -- it writes no further exchange word -- and raises the pose's `dirty` word; part 0, whose sums decide the solve
// and which alone stores the pose, reports status 2 ("a part never arrived") when IT ran out of patience.  The rescue launch that follows
// every split launch on the same stream (lc_pnp.hip: lc_pnp_lm_split_rescue_kernel; all parts have ended by then, there are no stragglers)
// re-zeroes the region of a dirty pose -- tickets and epoch start over, so whatever height the abandoned launch left behind cannot be taken
// for a current word -- and re-solves a pose of status 2 with ONE workgroup that plays the G parts one after the other: the same per-thread
// shares, the same wave / workgroup / part order of every sum (block_sum_parts_serial below), hence the bits the parts would have produced.
// So co-residency (pnp_split_parts() sizes grids to at most one workgroup per CU) is what makes the split form FAST; other streams, other
// processes or a CU mask taking compute units away make it slow, never wrong and never different.
// Tail of a pose's region (its last 128 bytes): word 0 epoch, word 1 dirty, word 2 rescues so far (diagnostics: tests read it).
constexpr int kSplitMaxParts = 8;
#ifndef LC_SPLIT_WAIT_US
#define LC_SPLIT_WAIT_US 2000
#endif
constexpr unsigned long long kSplitWaitTicks = 100ull * LC_SPLIT_WAIT_US;
// One unsuccessful poll more: true when the wait's budget is spent.  `t0` = 0 until the first clock read (the 16th unsuccessful poll).
__device__ __forceinline__ bool split_wait_expired(int polls, unsigned long long& t0) {
    if ((polls & 15) != 15) return false;
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (t0 == 0) { t0 = now; return false; }
    return now - t0 > kSplitWaitTicks;
}
constexpr int kSplitRowWords = 64;                                                             // 32 totals x 2 words
constexpr size_t kSplitPoseBytes = 2 * kSplitMaxParts * kSplitRowWords * sizeof(unsigned long long) + 128;  // two rows per part + the tail's line
constexpr int kSplitLdsDoubles = 128 + 64 + 2;
constexpr int kSplitSerialLdsDoubles = 128 + 32 * kSplitMaxParts + 32;  // block_sum_parts_serial: wave rows, one row per part, the totals
struct SplitSum {
    unsigned long long* exch;  // this pose's [2][kSplitMaxParts][64] words
    unsigned* epoch;           // this pose's tail: epoch, dirty, rescues
    int G, part;
    unsigned base;             // *epoch on entry
    unsigned seq;              // sums finished in this launch
    bool timed_out;
};
__device__ __forceinline__ unsigned* split_tail(void* region, size_t region_bytes) {
    return reinterpret_cast<unsigned*>(static_cast<char*>(region) + region_bytes - 128);
}
__device__ __forceinline__ SplitSum split_sum_enter(void* pose_region, int G, int part) {
    unsigned* epoch = split_tail(pose_region, kSplitPoseBytes);
    return SplitSum{reinterpret_cast<unsigned long long*>(pose_region), epoch, G, part, xcd_load(epoch), 0u, false};
}
// one thread of every part, after its last sum: part 0 of a launch that met every time moves the epoch on; a part that gave up marks the region
__device__ __forceinline__ void split_sum_leave(const SplitSum& sx) {
    if (sx.timed_out) xcd_store(sx.epoch + 1, 1u);
    else if (sx.part == 0 && sx.seq > 0) xcd_store(sx.epoch, sx.base + sx.seq);
}
// The rescue launch's entry, all threads of the workgroup that owns the region: false = nothing to do here.  A dirty region is zeroed
// (every word, the tail's rescue counter excepted); `redo` = the unit has to be computed again by this workgroup.
__device__ __forceinline__ bool split_rescue_enter(void* region, size_t region_bytes, bool redo, int tid, int threads) {
    unsigned* tail = split_tail(region, region_bytes);
    const bool dirty = xcd_load(tail + 1) != 0u;
    if (!dirty && !redo) return false;
    __syncthreads();  // every thread has read the tail before it is rewritten
    if (dirty) {
        unsigned long long* w = static_cast<unsigned long long*>(region);
        for (size_t i = tid; i < (region_bytes - 128) / 8; i += threads) xcd_store(w + i, 0ull);
        if (tid < 2) xcd_store(tail + tid, 0u);
    }
    if (redo && tid == 0) xcd_store(tail + 2, xcd_load(tail + 2) + 1u);
    return redo;
}

template <int K>
__device__ __forceinline__ void block_sum_split(double (&v)[K], double* lds, int tid, int& phase, SplitSum& sx) {
    static_assert(K <= 32, "one reduce-scatter of 32");
    const int lane = tid & 63, wave = tid >> 6;
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < K ? v[k] : 0.0;
    wave_reduce_scatter16<32>(w, lane);
    double* part = lds;                      // [4][32]
    double* tot = lds + 128 + 32 * phase;    // [2][32]
    int* arrived_ok = reinterpret_cast<int*>(lds + 192);
    phase ^= 1;
    if (sx.timed_out) {  // (uniform) this part has given up: it stays silent, its caller is on its way out
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = 0.0;
        return;
    }
    if ((lane & 3) == 0) *reinterpret_cast<double2*>(part + 32 * wave + scatter16_base(lane, 2)) = make_double2(w[0], w[1]);
    __syncthreads();
    if (tid < 32) {
        const unsigned long long ticket = (unsigned long long)(sx.base + sx.seq + 1u) << 32;
        unsigned long long* rows = sx.exch + (size_t)(sx.seq & 1u) * (kSplitMaxParts * kSplitRowWords) + 2 * tid;
        {
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, ((part[tid] + part[32 + tid]) + part[64 + tid]) + part[96 + tid]);
            unsigned long long* mine = rows + sx.part * kSplitRowWords;
            xcd_store(mine, ticket | (bits & 0xFFFFFFFFull));
            xcd_store(mine + 1, ticket | (bits >> 32));
        }
        // all G rows requested at once, again until every word carries the ticket (a row that is already there costs one read)
        unsigned long long lo[kSplitMaxParts], hi[kSplitMaxParts];
        bool ok = true;
        unsigned long long t0 = 0;
        for (int polls = 0;; ++polls) {
#pragma unroll
            for (int g = 0; g < kSplitMaxParts; ++g)
                if (g < sx.G) {
                    lo[g] = xcd_load(rows + g * kSplitRowWords);
                    hi[g] = xcd_load(rows + g * kSplitRowWords + 1);
                }
            unsigned long long late = 0;
#pragma unroll
            for (int g = 0; g < kSplitMaxParts; ++g)
                if (g < sx.G) late |= ((lo[g] ^ ticket) | (hi[g] ^ ticket)) >> 32;
            if (late == 0) break;
            if (split_wait_expired(polls, t0)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < kSplitMaxParts; ++g)  // in part order
            if (g < sx.G) s += __builtin_bit_cast(double, (hi[g] << 32) | (lo[g] & 0xFFFFFFFFull));
        tot[tid] = s;
        if (!ok) *arrived_ok = 0;  // (set to 1 by the workgroup before its first sum)
    }
    __syncthreads();
    if (!*arrived_ok) sx.timed_out = true;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = tot[k];
    ++sx.seq;
}

// block_sum_split played by ONE workgroup for the rescue launch: called once per part g = 0 .. G-1 with the values of that part's share
// (the same threads own the same correspondences as in workgroup g of the split launch), it forms the part's 32 totals exactly as
// block_sum_split does -- wave reduce-scatter, the four wave rows added in wave order -- and keeps them in LDS row g; the call for the last
// part adds the G rows in part order and hands every thread the totals.  lds: kSplitSerialLdsDoubles.
template <int K>
__device__ __forceinline__ void block_sum_parts_serial(double (&v)[K], double* lds, int tid, int g, int G) {
    static_assert(K <= 32, "one reduce-scatter of 32");
    const int lane = tid & 63, wave = tid >> 6;
    double w[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) w[k] = k < K ? v[k] : 0.0;
    wave_reduce_scatter16<32>(w, lane);
    double* part = lds;                               // [4][32]
    double* rows = lds + 128;                         // [G][32]
    double* tot = lds + 128 + 32 * kSplitMaxParts;    // [32]
    if ((lane & 3) == 0) *reinterpret_cast<double2*>(part + 32 * wave + scatter16_base(lane, 2)) = make_double2(w[0], w[1]);
    __syncthreads();
    if (tid < 32) {
        rows[32 * g + tid] = ((part[tid] + part[32 + tid]) + part[64 + tid]) + part[96 + tid];
        if (g == G - 1) {
            double s = 0.0;
            for (int h = 0; h < G; ++h) s += rows[32 * h + tid];  // in part order
            tot[tid] = s;
        }
    }
    __syncthreads();
    if (g == G - 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = tot[k];
    }
}

#ifndef LC_HORNER_ASM
#define LC_HORNER_ASM 1  // A/B switch (scripts/ubench/pnp_ab.py)
#endif
__device__ __forceinline__ double horner_step(double p, double z, double c) {
#if LC_HORNER_ASM
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "v"(c));
    return r;
#else
    return __builtin_fma(p, z, c);
#endif
}

// sin and cos of a rotation angle 0 <= x <~ 8 (angle-axis norms; the LM keeps them within a few pi): Cody-Waite reduction
// by pi/2 into [-pi/4, pi/4] and the fdlibm minimax polynomials (|error| < 1 ulp on that interval).  About a third of the
// instructions of the general sincos(), whose Payne-Hanek path for huge arguments is never needed here.
__device__ __forceinline__ void sincos_small(double x, double& s, double& c) {
    if (!(x < 1.0e4)) {  // wave-uniform and never taken in practice: keep full-range correctness anyway
        sincos(x, &s, &c);
        return;
    }
    const double kf = __builtin_rint(x * 0.63661977236758134308);  // x * 2/pi
    const int q = (int)kf;
    double r = __builtin_fma(-kf, 1.57079632673412561417e+00, x);   // pi/2 split in three parts (fdlibm pio2_1, _2, _3)
    r = __builtin_fma(-kf, 6.07710050650619224932e-11, r);
    r = __builtin_fma(-kf, 2.02226624879595063154e-21, r);
    const double z = r * r;
    // sin(r) ~ r + r^3 (S1 + z (S2 + ... )),  cos(r) ~ 1 - z/2 + z^2 (C1 + z (C2 + ...))   (fdlibm k_sin.c / k_cos.c)
    // Horner steps p z + C with the coefficient as the ADDEND: written as three-address v_fma_f64 with C in a register pair that stays
    // live across the LM loop -- the compiler's own form is `v_mov_b64 tmp, C; v_fmac_f64 tmp, z, p` (the accumulating two-address
    // encoding needs the addend in the destination), one extra VALU issue slot per coefficient and evaluation
    double ps = horner_step(1.58969099521155010221e-10, z, -2.50507602534068634195e-08);
    ps = horner_step(ps, z, 2.75573137070700676789e-06);
    ps = horner_step(ps, z, -1.98412698298579493134e-04);
    ps = horner_step(ps, z, 8.33333333332248946124e-03);
    ps = horner_step(ps, z, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(ps * z, r, r);
    double pc = horner_step(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09);
    pc = horner_step(pc, z, -2.75573143513906633035e-07);
    pc = horner_step(pc, z, 2.48015872894767294178e-05);
    pc = horner_step(pc, z, -1.38888888888741095749e-03);
    pc = horner_step(pc, z, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// atan(s / c) for s >= 0, c >= 0 (not both zero), result in [0, pi/2]: two argument halvings t -> t / (1 + sqrt(1 + t^2))
// (atan t = 2 atan of that) bring the ratio of the smaller to the larger below 0.199, where the odd Taylor series through z^21
// is exact to 2e-17 relative; measured against libm over 2e6 quaternions incl. tiny and half-turn rotations: <= 7e-16 relative (scripts/ubench/atan_accuracy.cpp).
// About a third of the instructions of libm's atan2, which sits on the start-up path of every PnP solve.
__device__ __forceinline__ double atan_ratio_pos(double s, double c) {
    const bool swap = s > c;
    const double a = swap ? c : s, b = swap ? s : c;
    double t = a * fast_rcp(b);  // in [0, 1]
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double h = fast_sqrt(__builtin_fma(t, t, 1.0));
        t = t * fast_rcp(1.0 + h);
    }
    const double w = t * t;
    double p = horner_step(-1.0 / 21.0, w, 1.0 / 19.0);
    p = horner_step(p, w, -1.0 / 17.0);
    p = horner_step(p, w, 1.0 / 15.0);
    p = horner_step(p, w, -1.0 / 13.0);
    p = horner_step(p, w, 1.0 / 11.0);
    p = horner_step(p, w, -1.0 / 9.0);
    p = horner_step(p, w, 1.0 / 7.0);
    p = horner_step(p, w, -1.0 / 5.0);
    p = horner_step(p, w, 1.0 / 3.0);
    const double at = 4.0 * __builtin_fma(-(p * w), t, t);  // 4 * (t - t^3 (1/3 - t^2/5 + ...))
    return swap ? 1.57079632679489661923 - at : at;
}

// A condition that is the same in every lane, as a SCALAR: lets hipcc branch with s_cbranch instead of saving/masking EXEC
// around code the whole wave takes or skips together (the LM loop's tests on wave-uniform doubles).
__device__ __forceinline__ bool uniform(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// index of (i,j), i<=j, in a packed upper-triangular 6x6 (21 entries, row-major)
__host__ __device__ constexpr int tri6(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

}  // namespace lc
