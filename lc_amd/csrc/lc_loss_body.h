// Device body of the fused linear-covariance pose loss (forward + analytic backward), one workgroup per sample.
// Included by lc_loss.hip (stand-alone kernel) and lc_fused.hip (loss + PnP in one launch).
//
// Replaces the ~1000-dispatch autograd graph of the reference's
//   lib/cov_mixed.py:100-150 Loss_cov_mixed  (+ lib/nll/pnp_auto.py, lib/nll/pnp_utils.py, lib/transforms/*)
// by ONE launch: every per-sample 2Nx6 Jacobian row lives in the registers of the lane that owns the point,
// the 6x6 normal equations / covariance algebra lives in LDS (one matrix entry per lane), and the only HBM
// traffic is the coalesced read of the (B,N,*) inputs and the write of loss + input gradients.
//
// Math (verified against the reference in oracle/lc_loss_oracle.py; symbols follow SURVEY.md 8a):
//   e = clamp(u - proj);  c = th(|e|, 3 mean|e|);  w = th(s, sqrt(4 mean(s^2 c)/(c+1e-6)))
//   H = sum w (J J^T + r Hess r);  S = H^-1 (or I if H is not SPD);  Mc = sum w^2 c J J^T;  v = sum w e J
//   G = d(bbox corners)/d(pose);  P = mean_k sqrt(tr_k(G S G^T)); C = mean_k sqrt(tr_k(G S Mc S G^T)); L = mean_k |G_k S v|
//   loss = log P + (C + L) / (2 P)
// and its hand-derived reverse mode (see profiles/r03/NOTES.md "LC-loss backward").
#pragma once
#include <type_traits>
#include "lc_common.h"
#include "lc_kernels.h"

// Diagnostic build only (-DLC_STAMPS, scripts/diag_stamps.py): s_memtime stamps per phase, written to p.aux which no
// other code reads in that build.  The shipped library never defines LC_STAMPS.
#ifdef LC_STAMPS
#define LC_STAMP(i)                                                                              \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (threadIdx.x == 0 && stamp_ok) reinterpret_cast<unsigned long long*>(p.aux)[(size_t)b * 20 + (i)] = t_; \
    } while (0)
#else
#define LC_STAMP(i) do {} while (0)
#endif

namespace lc {
namespace loss {

struct LossShared {
    double red[16][48];   // cross-wave partials of the 48-value reduction
    double small[16][4];  // cross-wave partials of the small reductions
    double A0[36];          // S = H^-1
    double sv[6];           // S*v
    double2 HM[24 * 6] __attribute__((aligned(16)));  // per bbox-Jacobian row j: pairs (h_j[a], m_j[a]), h_j = S g_j, m_j = S Mc h_j
    double pd[24], cd[24], dl[24], sq[24], isq[24];
    double Hbar[36], Psi[36], mu[6];
    int bad[4];  // [0],[1]: a non-positive diagonal in loss_cov_3d (P, C); [2]: H not SPD
};

// the one-workgroup loop form adds the per-tile partial rows of the canonical summation order
struct LossSharedLoop : LossShared {
    double p3[64][48];   // pass 3 (kLoopTiles rows)
};

struct PoseConst {
    double K[9];
    double R[9];   // the matrix the reference uses (two_s = 2/|q|, rotation_conversions.py:52)
    double Rt[9];  // proper rotation of the normalised quaternion
    double rho;
    double t[3];
};

__device__ __forceinline__ void quat_matrix(const double q[4], double two_s, double R[9]) {
    const double r = q[0], i = q[1], j = q[2], k = q[3];
    R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

struct Proj {
    double Xc[3];     // camera-frame point (unclamped)
    double proj[2];   // project_apply output (z clamped at 0.1, transforms.py:47-63)
    double zc;        // clamped depth of K*Xc
    double zpass;     // 1 if the clamp passes gradient (xf_z >= 0.1)
};

__device__ __forceinline__ Proj project(const PoseConst& pc, const double X[3]) {
    Proj o;
#pragma unroll
    for (int d = 0; d < 3; ++d) o.Xc[d] = pc.R[3 * d] * X[0] + pc.R[3 * d + 1] * X[1] + pc.R[3 * d + 2] * X[2] + pc.t[d];
    double xf[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) xf[d] = pc.K[3 * d] * o.Xc[0] + pc.K[3 * d + 1] * o.Xc[1] + pc.K[3 * d + 2] * o.Xc[2];
    o.zpass = xf[2] >= 0.1 ? 1.0 : 0.0;
    o.zc = xf[2] >= 0.1 ? xf[2] : 0.1;  // NaN propagates like torch.clamp? (NaN >= x false -> 0.1; inputs are finite)
    const double izc = fast_rcp(o.zc);  // zc >= 0.1
    o.proj[0] = xf[0] * izc;
    o.proj[1] = xf[1] * izc;
    return o;
}

// clamp_error (cov_mixed.py:16-24): cap the 2-vector length at max_len, identity gradient
__device__ __forceinline__ void clamp_err(const double u[2], const double proj[2], double max_len, double e[2]) {
    const double e0 = u[0] - proj[0], e1 = u[1] - proj[1];
    const double len = fast_sqrt(e0 * e0 + e1 * e1) + 1e-6;
    // f = (len - max)/len > 0  <=>  len > max (len >= 1e-6 > 0);  e = err - f err
    const double f = len > max_len ? (len - max_len) * fast_rcp(len) : 0.0;
    e[0] = e0 - f * e0;
    e[1] = e1 - f * e1;
}

// residual_with_jac6d (pnp_auto.py:13-56) at delta = 0, in closed form
struct PointJac {
    double J[2][6];
    double r[2];
    double iz, x0, y0;
    double M0[9];  // -R [X]x
};

__device__ __forceinline__ PointJac point_jac(const PoseConst& pc, const double X[3], const Proj& pr) {
    PointJac o;
    o.iz = 1.0 / pr.Xc[2];
    o.x0 = pr.Xc[0] * o.iz;
    o.y0 = pr.Xc[1] * o.iz;
    // M0 = R * [X]x^T,  [X]x^T = [[0, X2, -X1], [-X2, 0, X0], [X1, -X0, 0]]
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double r0 = pc.R[3 * d], r1 = pc.R[3 * d + 1], r2 = pc.R[3 * d + 2];
        o.M0[3 * d + 0] = -r1 * X[2] + r2 * X[1];
        o.M0[3 * d + 1] = r0 * X[2] - r2 * X[0];
        o.M0[3 * d + 2] = -r0 * X[1] + r1 * X[0];
    }
    double Ju[2][6];  // d uv0 / d delta = iz (T_a - uv0_a T_2), T = [M0 | I]
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        Ju[0][l] = o.iz * (o.M0[l] - o.x0 * o.M0[6 + l]);
        Ju[1][l] = o.iz * (o.M0[3 + l] - o.y0 * o.M0[6 + l]);
    }
    Ju[0][3] = o.iz; Ju[0][4] = 0; Ju[0][5] = -o.iz * o.x0;
    Ju[1][3] = 0; Ju[1][4] = o.iz; Ju[1][5] = -o.iz * o.y0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double k0 = pc.K[3 * c], k1 = pc.K[3 * c + 1];
#pragma unroll
        for (int l = 0; l < 6; ++l) o.J[c][l] = k0 * Ju[0][l] + k1 * Ju[1][l];
        o.r[c] = k0 * o.x0 + k1 * o.y0 + pc.K[3 * c + 2] - pr.proj[c];
    }
    return o;
}

// Workgroup-wide ordering point of the LDS hand-offs.  One-wave workgroups (N <= 64: the metric's shape) need neither a barrier nor
// a drain of the LDS queue -- a wave's DS operations execute in program order (lc_common.h: block_sum_sync).
__device__ __forceinline__ void wg_sync(int nw) {
    if (nw == 1 && !LC_WAVE_SYNC) wave_sync();
    else __syncthreads();
}

template <int K>
__device__ __forceinline__ void block_allreduce_small(double (&v)[K], double (*scratch)[4], int lane, int wave, int nw) {
    wave_allreduce<K>(v);
    if (nw > 1) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i) scratch[wave][i] = v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; ++i) {
            double s = 0;
            for (int w = 0; w < nw; ++w) s += scratch[w][i];
            v[i] = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Canonical summation order of the per-sample reductions (the same in every launch form, so a sample's results do not depend
// on how many workgroups share it): points are cut into TILES of 64 consecutive correspondences.  The 48 sums of pass 3: one
// wavefront reduces a tile with the cross-lane tree (wave_reduce_scatter16; lanes beyond N contribute exact zeros) and the tile
// partials are added sequentially in tile order.  The two small sums of passes 1-2: see the quarters at pass 1.
//   REG  (N <= 256): wave w of the workgroup is tile w, partials meet in LDS;
//   GRID (tiled form, N > 256): a 256-thread workgroup per 4, 8 or 16 tiles; the pass-3 partials of a sample meet in a
//        global workspace, ONE arrive-and-wait hand-off on a per-sample counter (below);
//   loop (!REG, one 256-thread workgroup per sample, for batches that fill the chip anyway): wave q walks quarter q of the tiles,
//        partials in LDS (up to kLoopTiles tiles; beyond that the per-thread accumulation of round 2 is kept).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTile = 64;
constexpr int kLoopTiles = 64;  // N <= 4096 in the canonical one-workgroup form

// Workspace of the tiled form (caller-provided, zero-initialised once, left zeroed by every launch):
//   header: [0] ticket, [1] samples done, [2] spin time-outs (diagnostic), [3] pad;  then 2 counters per sample (tiles arrived at
//   the hand-off, tiles finished);  then per sample and tile the 48 partial sums of pass 3.
constexpr int kGridRow = 48;
__host__ __device__ constexpr size_t grid_counter_words(int B) { return 4 + 2 * (size_t)B; }
__host__ __device__ constexpr size_t grid_rows_offset_bytes(int B) { return ((grid_counter_words(B) * 4 + 255) / 256) * 256; }
__host__ __device__ constexpr size_t grid_workspace_bytes(int B, int T) {
    return grid_rows_offset_bytes(B) + (size_t)B * T * kGridRow * sizeof(double);
}

struct GridCtx {
    unsigned* head;   // workspace header
    unsigned* ctr;    // this sample's two counters
    double* rows;     // this sample's (T, kGridRow) partials
    int T, S, TS, slice;  // tiles of the sample; workgroups sharing it; tiles per workgroup (4, 8 or 16); this workgroup's index among them
    int timed_out;
};

#ifndef LC_GRID_SPIN_LIMIT
#define LC_GRID_SPIN_LIMIT (1u << 22)  // ~seconds: a lost sibling (it cannot happen: tickets, below) must not hang the GPU
#endif
#ifndef LC_GRID_SLEEP
#define LC_GRID_SLEEP 2  // s_sleep argument between two polls of the arrival counter (64 cycles each)
#endif

// Hand-off between the S workgroups of a sample: the workgroup's tile partials have been stored with grid_store(); afterwards
// every sibling's can be read with grid_load().  No cache maintenance: an agent-scope release / acquire FENCE costs a write-back /
// invalidate of the whole L2 of the XCD per hand-off and workgroup (measured: the launch then scales with the number of
// workgroups, ~125 ns each); instead the few values exchanged are written through and read around the non-coherent caches
// (agent-scope relaxed atomics = sc1 accesses), ordered by the wave's own `s_waitcnt` before the arrival is counted.
// Deadlock-free without any co-residency assumption: workgroups take their (sample, tile) from a ticket counter when they START,
// so the tickets handed out are always a prefix 0..t-1 of the grid and every sample whose T tickets are inside the prefix has all
// its workgroups running; only the last, incomplete sample of the prefix waits for tickets not yet handed out, holds fewer than S
// workgroup slots, and gets its siblings as soon as any earlier sample retires.
#ifndef LC_GRID_FENCES
#define LC_GRID_FENCES 0  // A/B switch (scripts/ubench/tiled_loss.py): 1 = plain accesses + agent-scope fences
#endif
__device__ __forceinline__ void grid_store(double* q, double v) {
#if LC_GRID_FENCES
    *q = v;
#else
    __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ double grid_load(const double* q) {
#if LC_GRID_FENCES
    return *q;
#else
    return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void grid_arrive_wait(GridCtx& g, int tid) {
#if LC_GRID_FENCES
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
#endif
    __syncthreads();  // ... and so have the other waves' of the workgroup
    if (tid == 0) {
        unsigned* c = g.ctr;
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.S) {
            __builtin_amdgcn_s_sleep(LC_GRID_SLEEP);
            if (++spins > LC_GRID_SPIN_LIMIT) {
                __hip_atomic_fetch_add(g.head + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g.timed_out = 1;
                break;
            }
        }
    }
    __syncthreads();  // no wave loads a sibling's row before the poll has matched
#if LC_GRID_FENCES
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}

// sum over the tiles of column `col` of the sample's partial rows, in tile order
__device__ __forceinline__ double grid_column_sum(const GridCtx& g, int col) {
    double s = 0;
    const double* q = g.rows + col;
    int t = 0;
    for (; t + 4 <= g.T; t += 4) {  // four loads in flight, added in order
        const double a0 = grid_load(q + (size_t)t * kGridRow), a1 = grid_load(q + (size_t)(t + 1) * kGridRow),
                     a2 = grid_load(q + (size_t)(t + 2) * kGridRow), a3 = grid_load(q + (size_t)(t + 3) * kGridRow);
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; t < g.T; ++t) s += grid_load(q + (size_t)t * kGridRow);
    return s;
}

struct Pt {
    double X[3], u[2], s[2], vld;
};

__device__ __forceinline__ Pt load_pt(const LossParams& p, size_t base, int n) {
    Pt o;
    const float* X = p.pts3d + (base + n) * 3;
    const float2 u = *reinterpret_cast<const float2*>(p.pts2d + (base + n) * 2);
    const float2 s = *reinterpret_cast<const float2*>(p.inv_std + (base + n) * 2);
    o.X[0] = X[0]; o.X[1] = X[1]; o.X[2] = X[2];
    o.u[0] = u.x; o.u[1] = u.y;
    o.s[0] = s.x; o.s[1] = s.y;
    o.vld = p.valid ? (double)p.valid[base + n] : 1.0;
    return o;
}

// one correspondence as it sits in HBM: the walk below keeps a few of these per thread in registers across the passes
struct RawPt {
    float X[3];
    float2 u, s;
    float vld;
};
__device__ __forceinline__ RawPt load_raw_pt(const LossParams& p, size_t base, int n) {
    RawPt o;
    const float* X = p.pts3d + (base + n) * 3;
    o.u = *reinterpret_cast<const float2*>(p.pts2d + (base + n) * 2);
    o.s = *reinterpret_cast<const float2*>(p.inv_std + (base + n) * 2);
    o.X[0] = X[0]; o.X[1] = X[1]; o.X[2] = X[2];
    o.vld = p.valid ? p.valid[base + n] : 1.f;
    return o;
}
__device__ __forceinline__ Pt to_pt(const RawPt& r) {
    Pt o;
    o.X[0] = r.X[0]; o.X[1] = r.X[1]; o.X[2] = r.X[2];
    o.u[0] = r.u.x; o.u[1] = r.u.y;
    o.s[0] = r.s.x; o.s[1] = r.s.y;
    o.vld = r.vld;
    return o;
}

// REG (N <= 256): the workgroup has one thread per correspondence; raw inputs and clamped error stay in registers.
// !REG (N > 256), 256 threads, every pass is a WALK over tiles, four tiles per round, all loads of a round in flight together:
//   the two small sums walk the STATS range of the wave = quarter `wave` of ALL tiles of the sample (the canonical order above);
//   pass 3 and the backward pass walk the wave's OWN range:
//     loop form  (one workgroup per sample): own range = stats range;
//     tiled form (GRID: a sample shared by S workgroups, each owning TS = 4, 8 or 16 consecutive tiles): own range = quarter `wave`
//                of the workgroup's slice; every workgroup of the sample repeats the stats walk over the whole sample (two cheap
//                passes over <= 115 KB that sit in L2, instead of two more hand-offs); the 48 sums of pass 3 meet the sibling
//                workgroups' in the workspace (ONE hand-off).
//   A range of at most four tiles per wave is loaded ONCE (raw floats + clamped errors in registers) and serves all passes.
// COV2D: covariance of the projected bbox corners (cov_mixed.py:125-127) instead of the 3D ones (every reference call site).
// SH: LossShared (REG) or LossSharedLoop (!REG)
template <bool C, class A, class B>
__device__ __forceinline__ auto& pick_ref(A& a, B& b) {
    if constexpr (C) return a;
    else return b;
}

template <bool REG, bool COV2D = false, bool GRID = false, typename SH = LossShared>
__device__ __forceinline__ void sample(const LossParams& p, const int b, SH& sh, GridCtx* gc = nullptr) {
    static_assert(!GRID || !REG, "the tiled form is a walk form");
    constexpr bool WALK = !REG;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    const int N = p.N;
    const size_t base = (size_t)b * N;
    const int T = (N + kTile - 1) / kTile;
    const int my_pt = tid;  // REG: the correspondence this thread owns
    // loop form: canonical tile order while the partials fit the LDS rows; beyond that (N > 4096) per-thread accumulation
    [[maybe_unused]] const bool tiles = !REG && (GRID || T <= kLoopTiles);
    [[maybe_unused]] const bool stamp_ok = !GRID || gc->slice == 0;  // diagnostic build: one workgroup per sample writes the stamps

    LC_STAMP(0);
    constexpr int kRound = 4;
    // stats range (sq0, sq1) and own range (h0, h1) of this wave; WALK forms run four waves
    [[maybe_unused]] const int sq0 = (T * wave) >> 2, sq1 = (T * (wave + 1)) >> 2;
    [[maybe_unused]] int h0 = sq0, h1 = sq1;
    if constexpr (GRID) {
        const int s0 = gc->slice * gc->TS, len = min(gc->TS, T - s0);
        h0 = s0 + ((len * wave) >> 2);
        h1 = s0 + ((len * (wave + 1)) >> 2);
    }
    [[maybe_unused]] const bool cachedS = T <= 4 * kRound;               // every stats quarter fits one round
    [[maybe_unused]] const bool cachedH = GRID ? true : cachedS;         // (the launcher keeps TS <= 16)
    [[maybe_unused]] RawPt raw[kRound];        // stats range, while cachedS
    [[maybe_unused]] double ce[kRound][2];     // ... and its clamped errors
    [[maybe_unused]] RawPt raw_own[GRID ? kRound : 1];
    [[maybe_unused]] double ce_own[GRID ? kRound : 1][2];
    auto& rawH = pick_ref<GRID>(raw_own, raw);  // own range: its own registers in the tiled form, the stats range's in the loop form
    auto& ceH = pick_ref<GRID>(ce_own, ce);
    // branch-free on purpose: indices are clamped into the range / the sample, a lane without a correspondence carries a copy
    // with vld = 0 (exact zeros in the small sums), so the four tiles of a round are one basic block the scheduler can interleave
    [[maybe_unused]] auto load_round = [&](auto& rw, int t0, int t1, int r) {
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            const int t = t0 + r * kRound + k, n = t * kTile + lane;
            const bool live = t < t1 && n < N;
            rw[k] = load_raw_pt(p, base, live ? n : N - 1);
            if (!live) rw[k].vld = 0.f;
        }
    };
    // heavy(raw, ce, t, n, live): once per tile of the range, in tile order; live = this lane has a correspondence in the tile.
    // The body is instantiated ONCE: the loop over the round's tiles is not unrolled, the cached tiles ROTATE through slot 0
    // instead (a full turn after kRound steps: 48 register moves per tile against a body of thousands of cycles) -- four copies of
    // the pass-3 / backward bodies next to two caches do not fit the register file.
    [[maybe_unused]] auto walk = [&](auto& rw, auto& cw, int t0, int t1, bool cch, auto&& heavy) {
        if (t1 - t0 == 1) {  // the usual tiled shape (four tiles per workgroup): one tile per wave, nothing to rotate
            if (!cch) load_round(rw, t0, t1, 0);
            heavy(rw[0], cw[0], t0, t0 * kTile + lane, t0 * kTile + lane < N);
            return;
        }
        for (int r = 0; r < (t1 - t0 + kRound - 1) / kRound; ++r) {
            if (!cch) load_round(rw, t0, t1, r);
#pragma nounroll
            for (int k = 0; k < kRound; ++k) {
                const int t = t0 + r * kRound + k, n = t * kTile + lane;
                if (t < t1) heavy(rw[0], cw[0], t, n, n < N);
                const RawPt r0 = rw[0];
                const double c0 = cw[0][0], c1 = cw[0][1];
#pragma unroll
                for (int j = 0; j + 1 < kRound; ++j) { rw[j] = rw[j + 1]; cw[j][0] = cw[j + 1][0]; cw[j][1] = cw[j + 1][1]; }
                rw[kRound - 1] = r0; cw[kRound - 1][0] = c0; cw[kRound - 1][1] = c1;
            }
        }
    };
    // light(k): the same for the two small sums, branch-free (tiles beyond the range are all-dead copies)
    [[maybe_unused]] auto walk_light = [&](auto& rw, int t0, int t1, bool cch, auto&& light) {
        for (int r = 0; r < (t1 - t0 + kRound - 1) / kRound; ++r) {
            if (!cch) load_round(rw, t0, t1, r);
#pragma unroll
            for (int k = 0; k < kRound; ++k) light(k);
        }
    };
    if constexpr (WALK) {  // first thing in the kernel: the pose set-up below runs in the loads' shadow
        if constexpr (GRID) load_round(rawH, h0, h1, 0);
        if (cachedS) load_round(raw, sq0, sq1, 0);
    }

    PoseConst pc;
    {
        const float* Kp = p.K + 9 * (size_t)b;
        const float* ps = p.pose + 7 * (size_t)b;
#pragma unroll
        for (int i = 0; i < 9; ++i) pc.K[i] = Kp[i];
        const double q[4] = {ps[0], ps[1], ps[2], ps[3]};
        double irho;
        fast_sqrt_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], pc.rho, irho);
        quat_matrix(q, 2.0 * irho, pc.R);
        quat_matrix(q, 2.0 * irho * irho, pc.Rt);
        pc.t[0] = ps[4]; pc.t[1] = ps[5]; pc.t[2] = ps[6];
    }
    if (tid < 2) sh.bad[tid] = 0;
    // bbox corner of this lane's Jacobian row, fetched now so its HBM latency hides under passes 1-3
    // (loaded by EVERY lane, the index clamped, and kept as the floats that arrive: written as `if (tid < grows) { double = load; }` the
    // conversion sat next to the load inside the branch and the compiler waited for it there -- a memory round trip at the head of every
    // sample BEFORE the correspondences were even requested, the opposite of the intent)
    constexpr int gdim = COV2D ? 2 : 3;  // rows per bbox corner: projected (u,v) or transformed (x,y,z)   (cov_mixed.py:125-130)
    constexpr int grows = 8 * gdim;
    const float* const bb = p.bbox + ((size_t)b * 8 + (tid < grows ? tid : grows - 1) / gdim) * 3;
    const float bbxf = bb[0], bbyf = bb[1], bbzf = bb[2];

    const double max_len = p.max_err_len, rel_thresh = p.rel_thresh, w_e_thresh = p.w_e_thresh;

    Pt rp;          // REG only
    double re[2];   // REG only: clamped error
    bool active = false;
    if constexpr (REG) {
        active = my_pt < N;
        if (active) rp = load_pt(p, base, my_pt);
    }

    LC_STAMP(1);
    // ---------------- pass 1: e, sum |e| (robust_weights_cov, cov_mixed.py:27-31) ----------------
    // The two small reductions (3 and 2 values) in their canonical order: the tiles are cut into four QUARTERS
    // [T q / 4, T (q+1) / 4); a lane adds its points of a quarter sequentially (tile order), the wave tree reduces the quarter,
    // the four quarter totals are added in order.  REG: a wave is a tile is a quarter.  Walk forms: wave q walks quarter q.
    auto huber = [](double v, double d) { return v > d ? d * (2 * v - d) : v * v; };
    auto clamped_error = [&](const Pt& pt, double e[2]) {
        const Proj pr = project(pc, pt.X);
        clamp_err(pt.u, pr.proj, max_len, e);
    };
    double s1[3] = {0, 0, 0};
    if constexpr (REG) {
        if (active) {
            clamped_error(rp, re);
            s1[0] = fabs(re[0]) * rp.vld; s1[1] = fabs(re[1]) * rp.vld; s1[2] = rp.vld;
        }
    } else {
        walk_light(raw, sq0, sq1, cachedS, [&](int k) {
            const Pt pt = to_pt(raw[k]);
            clamped_error(pt, ce[k]);
            s1[0] += fabs(ce[k][0]) * pt.vld; s1[1] += fabs(ce[k][1]) * pt.vld; s1[2] += pt.vld;
        });
        if constexpr (GRID) {  // the clamped errors of the own tiles, for pass 3 and the backward pass
#pragma unroll
            for (int k = 0; k < kRound; ++k) clamped_error(to_pt(rawH[k]), ceH[k]);
        }
    }
    block_allreduce_small<3>(s1, sh.small, lane, wave, nw);
    const double vcnt = p.valid ? s1[2] : (double)N;
    const double ivcnt = fast_rcp(vcnt);  // vcnt == 0 (no valid point) gives NaN like the reference's 0/0
    const double dlt_e[2] = {s1[0] * ivcnt * rel_thresh, s1[1] * ivcnt * rel_thresh};  // Huber knee of |e|

    LC_STAMP(2);
    // ---------------- pass 2: c, mean(s^2 c) (cov_mixed.py:32-36) ----------------
    double s2[2] = {0, 0};
    // e of a walked tile: from the registers while the range is cached, else re-derived (in double: a float copy would lose bits)
    [[maybe_unused]] auto tile_error = [&](bool cch, const double (&c2)[2], const Pt& pt, double e[2]) {
        if (cch) { e[0] = c2[0]; e[1] = c2[1]; }
        else clamped_error(pt, e);
    };
    if constexpr (WALK) {
        walk_light(raw, sq0, sq1, cachedS, [&](int k) {
            const Pt pt = to_pt(raw[k]);
            double e[2];
            tile_error(cachedS, ce[k], pt, e);
#pragma unroll
            for (int c = 0; c < 2; ++c) s2[c] += pt.s[c] * pt.s[c] * huber(fabs(e[c]), dlt_e[c]) * pt.vld;
        });
    } else {
        if (active) {
#pragma unroll
            for (int c = 0; c < 2; ++c) s2[c] = rp.s[c] * rp.s[c] * huber(fabs(re[c]), dlt_e[c]) * rp.vld;
        }
    }
    block_allreduce_small<2>(s2, sh.small, lane, wave, nw);
    const double mwe[2] = {s2[0] * ivcnt * w_e_thresh, s2[1] * ivcnt * w_e_thresh};

    LC_STAMP(3);
    // ---------------- pass 3: accumulate H (21) | Mc (21) | v (6) ----------------
    double acc[48];
    // first == true: acc is written, not added to (a lane that owns one point needs no zero-fill and no extra add)
    auto accumulate = [&](const Pt& pt, const double e[2], auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const Proj pr = project(pc, pt.X);
        const PointJac pj = point_jac(pc, pt.X, pr);
        double w[2], cc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            cc[c] = huber(fabs(e[c]), dlt_e[c]);
            const double ds = fast_sqrt(mwe[c] * fast_rcp(cc[c] + 1e-6));
            w[c] = huber(pt.s[c], ds);
        }
        // second-order part  sum_c w_c r_c Hess(r_c)  (pnp_auto.py:59-83; exact 2nd derivative, see oracle)
        const double rho0 = pc.K[0] * w[0] * pj.r[0] + pc.K[3] * w[1] * pj.r[1];
        const double rho1 = pc.K[1] * w[0] * pj.r[0] + pc.K[4] * w[1] * pj.r[1];
        const double iz2 = pj.iz * pj.iz;
        const double qv[3] = {-rho0 * iz2, -rho1 * iz2, (rho0 * pj.x0 + rho1 * pj.y0) * iz2};
        double tq[6], t2[6];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            tq[l] = pj.M0[l] * qv[0] + pj.M0[3 + l] * qv[1] + pj.M0[6 + l] * qv[2];
            tq[3 + l] = qv[l];
            t2[l] = pj.M0[6 + l];
            t2[3 + l] = l == 2 ? 1.0 : 0.0;
        }
        const double pv[3] = {pj.iz * rho0, pj.iz * rho1, -pj.iz * (rho0 * pj.x0 + rho1 * pj.y0)};
        double pi[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) pi[l] = pc.R[l] * pv[0] + pc.R[3 + l] * pv[1] + pc.R[6 + l] * pv[2];
        const double piX = pi[0] * pt.X[0] + pi[1] * pt.X[1] + pi[2] * pt.X[2];
        const double m0 = w[0] * w[0] * cc[0], m1 = w[1] * w[1] * cc[1], we0 = w[0] * e[0], we1 = w[1] * e[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double wj0 = w[0] * pj.J[0][i], wj1 = w[1] * pj.J[1][i], mj0 = m0 * pj.J[0][i], mj1 = m1 * pj.J[1][i];
#pragma unroll
            for (int j = i; j < 6; ++j) {
                double h = t2[i] * tq[j] + tq[i] * t2[j];  // compile-time zeros of t2 fold away
                if (i < 3 && j < 3) h += 0.5 * (pi[i] * pt.X[j] + pt.X[i] * pi[j]) - (i == j ? piX : 0.0);
                const double hv = __builtin_fma(wj1, pj.J[1][j], __builtin_fma(wj0, pj.J[0][j], h));
                const double mv = __builtin_fma(mj1, pj.J[1][j], mj0 * pj.J[0][j]);
                acc[tri6(i, j)] = FIRST ? hv : acc[tri6(i, j)] + hv;
                acc[21 + tri6(i, j)] = FIRST ? mv : acc[21 + tri6(i, j)] + mv;
            }
            const double vv = __builtin_fma(we1, pj.J[1][i], we0 * pj.J[0][i]);
            acc[42 + i] = FIRST ? vv : acc[42 + i] + vv;
        }
    };
    if constexpr (REG) {
        if (active) {
            accumulate(rp, re, std::true_type{});
        } else {
#pragma unroll
            for (int i = 0; i < 48; ++i) acc[i] = 0;
        }
    } else {
        if (tiles) {  // a tile's 48 sums by the cross-lane tree, one row per tile: LDS (loop form) or the sample's workspace rows (tiled form)
            walk(rawH, ceH, h0, h1, cachedH, [&](const RawPt& rw, const double (&cw)[2], int t, int, bool live) {
                if (live) {
                    const Pt pt = to_pt(rw);
                    double e[2];
                    tile_error(cachedH, cw, pt, e);
                    accumulate(pt, e, std::true_type{});
                } else {
#pragma unroll
                    for (int i = 0; i < 48; ++i) acc[i] = 0;
                }
                wave_reduce_scatter16<48>(acc, lane);
                if ((lane & 3) == 0) {
                    const int bs = scatter16_base(lane, 3);
                    if constexpr (GRID) {
                        double* row = gc->rows + (size_t)t * kGridRow + bs;
#pragma unroll
                        for (int i = 0; i < 3; ++i) grid_store(row + i, acc[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i) sh.p3[t][bs + i] = acc[i];
                    }
                }
            });
        } else {
#pragma unroll
            for (int i = 0; i < 48; ++i) acc[i] = 0;
            for (int n = tid; n < N; n += nthr) {
                const Pt pt = load_pt(p, base, n);
                double e[2];
                clamped_error(pt, e);
                accumulate(pt, e, std::false_type{});
            }
        }
    }
    LC_STAMP(4);
    if constexpr (GRID) {  // totals in tile order once every sibling workgroup's rows have arrived
        grid_arrive_wait(*gc, tid);
        if (T <= kLoopTiles) {
            // all T rows in ONE round trip: every thread fetches a few values around the caches into LDS, 48 threads add the columns in tile order
            // (up to eight requests per thread issued before the first is stored, no branch between them: the plain loop compiled to four in
            // flight and then one round trip per remaining iteration -- three dependent round trips for zlmo's 29 tiles)
            const int total = T * kGridRow;
            for (int i0 = tid; i0 < total; i0 += 8 * nthr) {
                double tmp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tmp[u] = grid_load(gc->rows + min(i0 + u * nthr, total - 1));
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * nthr < total) (&sh.p3[0][0])[i0 + u * nthr] = tmp[u];
            }
            __syncthreads();
            if (tid < 48) {
                double s = 0;
                for (int t = 0; t < T; ++t) s += sh.p3[t][tid];
                sh.red[0][tid] = s;
            }
        } else if (tid < 48) {
            sh.red[0][tid] = grid_column_sum(*gc, tid);
        }
        __syncthreads();
    } else if (!REG && tiles) {
        if constexpr (!REG) {
            __syncthreads();
            if (tid < 48) {
                double s = 0;
                for (int t = 0; t < T; ++t) s += sh.p3[t][tid];
                sh.red[0][tid] = s;
            }
            __syncthreads();
        }
    } else {
        wave_reduce_scatter16<48>(acc, lane);
        if ((lane & 3) == 0) {
            const int bs = scatter16_base(lane, 3);
#pragma unroll
            for (int i = 0; i < 3; ++i) sh.red[wave][bs + i] = acc[i];
        }
        wg_sync(nw);
        if (nw > 1) {  // combine the waves' partials into sh.red[0] (wave = tile: the canonical order)
            if (tid < 48) {
                double s = 0;
                for (int w = 0; w < nw; ++w) s += sh.red[w][tid];
                sh.red[0][tid] = s;
            }
            wg_sync(nw);
        }
    }
    // packed totals (upper triangles, so H and Mc are symmetric by construction: make_sure_symmetric, pnp_utils.py:134-137)
    const double* Hp = &sh.red[0][0];    // H  (21)
    const double* Mp = &sh.red[0][21];   // Mc (21)
    const double* vp = &sh.red[0][42];   // v  (6)

    LC_STAMP(5);
    // ---------------- serial section, one matrix entry per lane ----------------
    const int mi = (tid % 36) / 6, mj = tid % 6;  // (row, col) of this lane's entry; lanes >= 36 shadow lanes 0..27
    // S = H^-1 by Gauss-Jordan sweeps held in REGISTERS of wave 0 (one entry per lane; pivot row / column entries come
    // through ds_bpermute, no LDS round trip per sweep).  Pivots == squared Cholesky diagonal -> SPD test of safe_cholesky.
    if (wave == 0) {
        double cur = Hp[mi <= mj ? tri6(mi, mj) : tri6(mj, mi)];
        bool ok = true;
#pragma unroll
        for (int pz = 0; pz < 6; ++pz) {
            const double piv = shfl_f64(cur, 7 * pz);
            const double rowv = shfl_f64(cur, 6 * pz + mj);
            const double colv = shfl_f64(cur, 6 * mi + pz);
            ok = ok && (piv > 0);
            const double ip = fast_rcp(piv);
            const double t = rowv * ip;
            double out = cur - colv * t;
            if (mj == pz) out = -colv * ip;
            if (mi == pz) out = (mj == pz) ? ip : t;
            cur = out;
        }
        if (!ok) cur = (mi == mj) ? 1.0 : 0.0;  // make_sure_SPD (pnp_utils.py:140-157): H := I, no gradient into H
        if (tid < 36) sh.A0[tid] = cur;
        if (tid == 0) sh.bad[2] = ok ? 0 : 1;
    }
    wg_sync(nw);
    const bool spd = sh.bad[2] == 0;
    LC_STAMP(6);
    const double* S = sh.A0;
    // step A: lanes 0..23 own one row g_j = [ -rho (Rt[d,:] x b_k) | e_d ] of the bbox Jacobian (jac_update2alter,
    // cov_mixed.py:42-65):  h_j = S g_j;  diag(G S G^T)_j = g.h;  n_j = Mc h_j;  diag(G S Mc S G^T)_j = h.n;
    // (G S v)_j = h.v;  m_j = S n_j (so that Psi*Mc*S = sum cC_j h_j m_j^T in the reverse section).
    if (tid < grows) {
        const double bbx = bbxf, bby = bbyf, bbz = bbzf;  // the corner of this lane's row, requested at the kernel's start
        double g[6];
        if constexpr (!COV2D) {  // xform_3d (cov_mixed.py:73-75): rows g = [ -rho (Rt[d,:] x b) | e_d ]
            const int d = tid % 3;
            // static selects (a runtime-indexed pc.Rt[3*d] would push the whole PoseConst to scratch)
            const double r0 = d == 0 ? pc.Rt[0] : (d == 1 ? pc.Rt[3] : pc.Rt[6]);
            const double r1 = d == 0 ? pc.Rt[1] : (d == 1 ? pc.Rt[4] : pc.Rt[7]);
            const double r2 = d == 0 ? pc.Rt[2] : (d == 1 ? pc.Rt[5] : pc.Rt[8]);
            g[0] = -pc.rho * (r1 * bbz - r2 * bby);
            g[1] = -pc.rho * (r2 * bbx - r0 * bbz);
            g[2] = -pc.rho * (r0 * bby - r1 * bbx);
            g[3] = d == 0 ? 1.0 : 0.0; g[4] = d == 1 ? 1.0 : 0.0; g[5] = d == 2 ? 1.0 : 0.0;
        } else {  // xform_2d (cov_mixed.py:78-80): project_apply of the corner, chained with the 3D rows above
            const int a = tid % 2;
            const double bX[3] = {bbx, bby, bbz};
            const Proj pr = project(pc, bX);
            const double izc = fast_rcp(pr.zc);
            double pa[3];  // d proj_a / d Xc = (K[a,:] - zpass proj_a K[2,:]) / zc
#pragma unroll
            for (int l = 0; l < 3; ++l) pa[l] = ((a == 0 ? pc.K[l] : pc.K[3 + l]) - pr.zpass * pr.proj[a] * pc.K[6 + l]) * izc;
            // sum_d pa[d] * (-rho (Rt[d,:] x b)) = -rho ((pa^T Rt) x b)
            const double v0 = pa[0] * pc.Rt[0] + pa[1] * pc.Rt[3] + pa[2] * pc.Rt[6];
            const double v1 = pa[0] * pc.Rt[1] + pa[1] * pc.Rt[4] + pa[2] * pc.Rt[7];
            const double v2 = pa[0] * pc.Rt[2] + pa[1] * pc.Rt[5] + pa[2] * pc.Rt[8];
            g[0] = -pc.rho * (v1 * bbz - v2 * bby);
            g[1] = -pc.rho * (v2 * bbx - v0 * bbz);
            g[2] = -pc.rho * (v0 * bby - v1 * bbx);
            g[3] = pa[0]; g[4] = pa[1]; g[5] = pa[2];
        }
        double Sr[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) Sr[i] = S[i];
        double h[6], nn[6], pd = 0, dl = 0, cd = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double a2 = Sr[6 * i] * g[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) a2 = __builtin_fma(Sr[6 * i + k], g[k], a2);
            h[i] = a2;
            pd = __builtin_fma(g[i], a2, pd);
            dl = __builtin_fma(a2, vp[i], dl);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double a2 = Mp[tri6(0, i)] * h[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) a2 = __builtin_fma(Mp[i <= k ? tri6(i, k) : tri6(k, i)], h[k], a2);
            nn[i] = a2;
            cd = __builtin_fma(h[i], a2, cd);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double a2 = Sr[6 * i] * nn[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) a2 = __builtin_fma(Sr[6 * i + k], nn[k], a2);
            sh.HM[6 * tid + i] = make_double2(h[i], a2);
        }
        sh.pd[tid] = pd; sh.cd[tid] = cd; sh.dl[tid] = dl;
        if (!(pd > 0)) sh.bad[0] = 1;  // loss_cov_3d / loss_cov_2d 'good' (cov_mixed.py:83-97)
        if (!(cd > 0)) sh.bad[1] = 1;
    }
    wg_sync(nw);
    LC_STAMP(7);
    // step B: sqrt of the 8 corner traces (P, C) and the 8 corner norms (L) and their reciprocals; S v on six idle lanes
    if (tid < 24) {
        const int q = tid >> 3, k = tid & 7;
        const double* src = q == 0 ? sh.pd : (q == 1 ? sh.cd : sh.dl);
        const double a0 = src[gdim * k], a1 = src[gdim * k + 1], a2 = COV2D ? 0.0 : src[gdim * k + (COV2D ? 1 : 2)];
        double val, ival;
        fast_sqrt_rsqrt(q == 2 ? a0 * a0 + a1 * a1 + a2 * a2 : (sh.bad[q] ? 1.0 : a0 + a1 + a2), val, ival);
        sh.sq[tid] = val;
        sh.isq[tid] = val > 0 ? ival : 0.0;  // norm backward at 0 is 0 (torch.linalg.vector_norm)
    } else if (tid < 30) {
        const int a = tid - 24;
        double acc2 = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc2 += S[6 * a + k] * vp[k];
        sh.sv[a] = acc2;
    }
    wg_sync(nw);
    LC_STAMP(8);
    double sP[8], sC[8], sL[8], Pm = 0, Cm = 0, Lm = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sP[k] = sh.sq[k]; sC[k] = sh.sq[8 + k]; sL[k] = sh.sq[16 + k];
        Pm += sP[k]; Cm += sC[k]; Lm += sL[k];
    }
    Pm *= 0.125; Cm *= 0.125; Lm *= 0.125;
    const double iP = fast_rcp(Pm);
    const double loss = log(Pm) + 0.5 * (Cm + Lm) * iP;
    const double gout = p.grad_out ? (double)p.grad_out[b] : 1.0;
    const bool writer = !GRID || gc->slice == 0;  // tiled form: every workgroup of the sample computes the (identical) sample-level section, the first stores it
    if (tid == 0 && writer) {
        p.loss[b] = (float)loss;
#ifndef LC_STAMPS
        if (p.aux) {
            float* ax = p.aux + (size_t)b * kLossAuxStride;
            ax[0] = (float)Pm; ax[1] = (float)Cm; ax[2] = (float)Lm; ax[3] = spd ? 0.f : 1.f;
        }
#endif
    }
#ifndef LC_STAMPS
    if (p.aux && tid < 36 && writer) p.aux[(size_t)b * kLossAuxStride + 4 + tid] = (float)S[tid];
#endif
    if (p.d_pts2d == nullptr) return;  // forward only

    LC_STAMP(9);
    // ---------------- reverse mode of the serial section ----------------
    // With h_j = S g_j, m_j = S Mc h_j:  S Phi_P S = sum cP_j h h^T (PsiP),  Psi = S Phi_C S = sum cC_j h h^T,
    //   W = Psi Mc S = sum cC_j h m^T,  mu = S lambda = sum cL_j dl_j h_j,
    //   Hbar = -S Sbar S = -( PsiP + W + W^T + (mu sv^T + sv mu^T)/2 );  lane (a,b) forms its own entry of every term.
    const double aP = gout * (iP - 0.5 * (Cm + Lm) * iP * iP);
    const double aC = gout * 0.5 * iP;  // = dloss/dC = dloss/dL
    const bool badP = sh.bad[0] != 0, badC = sh.bad[1] != 0;
    if (tid < 36) {
        double fp = 0, fc = 0, fw = 0, la = 0, lb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double hh = 0, ww = 0, da = 0, db = 0;
#pragma unroll
            for (int d = 0; d < gdim; ++d) {
                const int j = gdim * k + d;
                const double2 xa = sh.HM[6 * j + mi], xb = sh.HM[6 * j + mj];
                const double dlj = sh.dl[j];
                hh = __builtin_fma(xa.x, xb.x, hh);
                ww = __builtin_fma(xa.x, xb.y, __builtin_fma(xb.x, xa.y, ww));  // W[a][b] + W[b][a]
                da = __builtin_fma(dlj, xa.x, da);
                db = __builtin_fma(dlj, xb.x, db);
            }
            fp = __builtin_fma(sh.isq[k], hh, fp);
            fc = __builtin_fma(sh.isq[8 + k], hh, fc);
            fw = __builtin_fma(sh.isq[8 + k], ww, fw);
            la = __builtin_fma(sh.isq[16 + k], da, la);
            lb = __builtin_fma(sh.isq[16 + k], db, lb);
        }
        const double psiP = badP ? 0.0 : fp * (aP * 0.0625);  // cP_k = aP / (16 sP_k)
        const double cC = badC ? 0.0 : aC * 0.0625;           // cC_k = aC / (16 sC_k)
        const double mua = la * (aC * 0.125), mub = lb * (aC * 0.125);  // cL_k = aC / (8 sL_k)
        const double hb = -(psiP + cC * fw + 0.5 * (mua * sh.sv[mj] + sh.sv[mi] * mub));
        sh.Psi[tid] = cC * fc;
        sh.Hbar[tid] = spd ? hb : 0.0;
        if (mj == 0) sh.mu[mi] = mua;
    }
    wg_sync(nw);

    LC_STAMP(10);
    // ---------------- pass 4: per-point gradients ----------------
    // symmetric 6x6 forms from registers: packed upper triangles with the off-diagonals doubled
    double Hs[21], Ps[21], mu[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
            const double f = i == j ? 1.0 : 2.0;
            Hs[tri6(i, j)] = f * sh.Hbar[6 * i + j];
            Ps[tri6(i, j)] = f * sh.Psi[6 * i + j];
        }
        mu[i] = sh.mu[i];
    }
    const double trHww = Hs[tri6(0, 0)] + Hs[tri6(1, 1)] + Hs[tri6(2, 2)];
    auto backward_point = [&](const Pt& pt, const double e[2], int n) {
        const Proj pr = project(pc, pt.X);
        const PointJac pj = point_jac(pc, pt.X, pr);
        double w[2], cc[2], dsk[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            cc[c] = huber(fabs(e[c]), dlt_e[c]);
            dsk[c] = fast_sqrt(mwe[c] * fast_rcp(cc[c] + 1e-6));
            w[c] = huber(pt.s[c], dsk[c]);
        }
        // h2 = Hbar * t2 with t2 = (M0[2,:], 0, 0, 1);  q3 = Hbar_ww X - tr(Hbar_ww) X   (Hs off-diagonals are doubled)
        double h2[6], q3[3];
        auto Hent = [&](int i, int j) { return i == j ? Hs[tri6(i, i)] : 0.5 * Hs[i < j ? tri6(i, j) : tri6(j, i)]; };
#pragma unroll
        for (int i = 0; i < 6; ++i) h2[i] = Hent(i, 0) * pj.M0[6] + Hent(i, 1) * pj.M0[7] + Hent(i, 2) * pj.M0[8] + Hent(i, 5);
#pragma unroll
        for (int i = 0; i < 3; ++i) q3[i] = Hent(i, 0) * pt.X[0] + Hent(i, 1) * pt.X[1] + Hent(i, 2) * pt.X[2] - trHww * pt.X[i];
        const double iz2 = pj.iz * pj.iz;
        double Bq[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double qa[3] = {a == 0 ? -iz2 : 0.0, a == 1 ? -iz2 : 0.0, (a == 0 ? pj.x0 : pj.y0) * iz2};
            double tqh = 0;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                tqh += (pj.M0[l] * qa[0] + pj.M0[3 + l] * qa[1] + pj.M0[6 + l] * qa[2]) * h2[l];
                tqh += qa[l] * h2[3 + l];
            }
            const double pa[3] = {a == 0 ? pj.iz : 0.0, a == 1 ? pj.iz : 0.0, -pj.iz * (a == 0 ? pj.x0 : pj.y0)};
            double piq = 0;
#pragma unroll
            for (int l = 0; l < 3; ++l) piq += (pc.R[l] * pa[0] + pc.R[3 + l] * pa[1] + pc.R[6 + l] * pa[2]) * q3[l];
            Bq[a] = 2.0 * tqh + piq;
        }
        double gs[2], ge[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double qH = 0, qPsi = 0, jm = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = i; j < 6; ++j) {
                    const double pp = pj.J[c][i] * pj.J[c][j];
                    qH += Hs[tri6(i, j)] * pp;
                    qPsi += Ps[tri6(i, j)] * pp;
                }
                jm += pj.J[c][i] * mu[i];
            }
            const double wbar = qH + 2.0 * w[c] * cc[c] * qPsi + e[c] * jm +
                                pj.r[c] * (pc.K[3 * c] * Bq[0] + pc.K[3 * c + 1] * Bq[1]);
            const double cbar = w[c] * w[c] * qPsi;
            gs[c] = wbar * (pt.s[c] > dsk[c] ? 2.0 * dsk[c] : 2.0 * pt.s[c]);
            const double a = fabs(e[c]);
            const double sgn = e[c] > 0 ? 1.0 : (e[c] < 0 ? -1.0 : 0.0);
            ge[c] = cbar * (a > dlt_e[c] ? 2.0 * dlt_e[c] : 2.0 * a) * sgn;
        }
        *reinterpret_cast<float2*>(p.d_pts2d + (base + n) * 2) = make_float2((float)ge[0], (float)ge[1]);
        *reinterpret_cast<float2*>(p.d_inv_std + (base + n) * 2) = make_float2((float)gs[0], (float)gs[1]);
        if (p.d_pts3d) {
            // err = u - proj: dX = -(d proj/d X)^T ge, d proj_a/dX = (KR[a,:] - zpass * proj_a KR[2,:]) / zc
            double gx[3];
            const double izc = fast_rcp(pr.zc);  // zc >= 0.1
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double kr0 = pc.K[0] * pc.R[l] + pc.K[1] * pc.R[3 + l] + pc.K[2] * pc.R[6 + l];
                const double kr1 = pc.K[3] * pc.R[l] + pc.K[4] * pc.R[3 + l] + pc.K[5] * pc.R[6 + l];
                const double kr2 = pc.K[6] * pc.R[l] + pc.K[7] * pc.R[3 + l] + pc.K[8] * pc.R[6 + l];
                gx[l] = -((kr0 - pr.zpass * pr.proj[0] * kr2) * ge[0] + (kr1 - pr.zpass * pr.proj[1] * kr2) * ge[1]) * izc;
            }
            float* o = p.d_pts3d + (base + n) * 3;
            o[0] = (float)gx[0]; o[1] = (float)gx[1]; o[2] = (float)gx[2];
        }
    };
    if constexpr (REG) {
        if (active) backward_point(rp, re, my_pt);
    } else {
        walk(rawH, ceH, h0, h1, cachedH, [&](const RawPt& rw, const double (&cw)[2], int, int n, bool live) {
            if (live) {
                const Pt pt = to_pt(rw);
                double e[2];
                tile_error(cachedH, cw, pt, e);
                backward_point(pt, e, n);
            }
        });
    }
    LC_STAMP(11);
}

}  // namespace loss
}  // namespace lc
