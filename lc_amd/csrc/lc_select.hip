// Test-time point selection for the dense heads (second half): per-sample mask / weight-quantile
// selection of the dense correspondences and their COMPACTION into padded per-sample lists + counts, on the device.
//
// Replaces test.py:39-45 (quantile_msk: torch.quantile over the summed weights), test.py:94-106 (the three
// dense_point_select modes, one nonzero() per sample = one host sync per sample, ragged Python lists, np.random padding)
// and the re-batching of those lists in cer_solver.py:67-87.  One workgroup per sample:
//   1. weight_i = inv_std[i,0] + inv_std[i,1]  (x seg_i in mode 2)
//   2. modes 1/2: radix select of the two order statistics around q*(n-1) from the weights staged in LDS, threshold =
//      torch.quantile's linear interpolation between them  (mode 2: q_b = 1 - (1-q) * mean(seg), test.py:102-103)
//   3. keep_i = seg_i | weight_i >= thr | (weight_i >= thr) & seg_i
//   4. order-preserving compaction (wave ballots + prefix over the waves) of pts2d, weights (optionally squared: the
//      inverse covariance the solver takes, test.py:92), pts3d and the source indices; count per sample
//   5. fewer than min_count survivors (and more than min_count candidates): pad with pseudo-random source indices, the
//      role np.random.choice plays in test.py:108-113 (seeded hash instead of the host RNG)
// The outputs feed lc_pnp_ransac_init_f32 / lc_pnp_lm_f32 directly through their `counts` argument.
// Two kernels over the same body (select_row): lc_dense_select_kernel reads the rows the dense front end wrote (ArraySource);
// lc_dense_frontend_select_kernel forms them itself from the network's maps (MapSource: front end + selection in one launch, the rows
// in between never exist).  The body is bound by barriers of 16 wavefronts and by LDS atomics on clustered keys: one barrier per radix
// pass (three histograms in rotation, every wavefront scans), wave-aggregated histogram adds, the thread's own entry held in registers.
#include <cfloat>

#include "lc_common.h"
#include "lc_dense_lse.h"
#include "lc_kernels.h"
#include "lc_select_rows.h"

#ifdef LC_SELECT_STAMPS  // diagnostic build (scripts/ubench/select_stamps.py): s_memtime stamps of the wide front end + selection's phases
namespace lc { namespace select_diag { __device__ unsigned long long g_stamp[10]; } }
#define LC_FS_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) ::lc::select_diag::g_stamp[i] = __builtin_amdgcn_s_memtime()
#else
#define LC_FS_STAMP(i)
#endif

namespace lc {
namespace {

constexpr int kThreads = 1024;
constexpr int kFusedSelectMaxPoints = 16384;  // lc_dense_frontend_select_f32: keys in up to 64 KB of the 160 KB LDS (128x128 maps at stride 1, zlmo's test-time shape); entries beyond the thread's first four are re-formed from the maps
constexpr int kWaves = kThreads / kWave;

// torch.lerp (aten/src/ATen/native/Lerp.h): the form that is exact at both ends
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
    return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.f - w);
}

// One count into an LDS histogram per lane that is `active`.  Weights cluster -- a quantile inside a mask zeroes every weight outside
// it, and the top byte of positive floats of similar size is the same -- so a plain atomicAdd per lane sends hundreds of adds to ONE
// address, which the LDS serialises.  Two rounds of leader aggregation (the lanes that share the first active lane's bin are counted
// by one add of their ballot's population) take the crowd off; whoever is left adds for itself.
// aggregate: the leader rounds pay on the TOP digit only (sign + high exponent bits: nearly every key in a handful of bins); on the lower
// digits the keys that are still in play spread over the bins and the two rounds were pure latency.  The leader's bin is fetched with
// v_readlane (its lane index is wave-uniform), not through the LDS crossbar: at 16 384 candidates per object the selection was bound by
// the number of LDS instructions it issued (key read + two permutes + atomics per key and pass).
__device__ __forceinline__ void hist_add(int* hist, unsigned bin, bool active, int lane, bool aggregate) {
    if (aggregate) {
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const unsigned long long act = __ballot(active);
            if (act == 0ull) return;  // wave-uniform
            const int leader = __ffsll((long long)act) - 1;
            const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
            const unsigned long long same = __ballot(active && bin == lb);
            if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
            active = active && bin != lb;
        }
    }
    if (active) atomicAdd(&hist[bin], 1);
}

// One entry of a row: image point, weights, model point, source index, mask bit.
struct Entry {
    float2 u, s;
    float X[3];
    int src;
    unsigned char g;
};

// Rows held in arrays (lc_dense_select_f32)
struct ArraySource {
    const SelectParams& p;
    size_t base;
    __device__ __forceinline__ Entry load(int i) const {
        Entry e;
        e.s = *reinterpret_cast<const float2*>(p.inv_std + (base + i) * 2);
        e.g = p.mask ? p.mask[base + i] : 0;
        e.u = *reinterpret_cast<const float2*>(p.pts2d + (base + i) * 2);
        for (int d = 0; d < 3; ++d) e.X[d] = p.pts3d[(base + i) * 3 + d];
        e.src = p.in_index ? p.in_index[base + i] : i;
        return e;
    }
};

// Rows that exist only as the network's maps (lc_dense_frontend_select_f32): entry n is sampled pixel n of the dense front end,
// formed with the front end's own arithmetic (lc_dense.hip: lc_dense_frontend_fwd_kernel)
template <typename T, typename TX>  // element type of the logit maps / of the xyz map (lc_map.h; TX = T, or float: decoded code planes)
struct MapSource {
    const T* lg;   // (2,H,W) weight logits of the sample
    const TX* xyz; // (3,H,W)
    const T* vis;  // (H,W) visibility logits or null
    float lse, scale, ns[3], vis_thresh;
    int HW, W, Wn, top, left, sample;
    __device__ __forceinline__ int pixel(int n, int& x, int& y) const {
        const int r = n / Wn;
        y = top + r * sample;
        x = left + (n - r * Wn) * sample;
        return y * W + x;
    }
    // the part that does not need the log-sum-exp (requested while it is being formed): logits in s, raw xyz in X
    // Loads ONLY (the visibility logit comes back raw in `v`): nothing in here waits for memory, so the requests of several entries --
    // each behind its own `if (index < N)` -- are all in flight together (with the sigmoid in here an entry was a round trip of its own)
    __device__ __forceinline__ Entry fetch(int n, float& v) const {
        int x, y;
        const int px = pixel(n, x, y);
        Entry e;
        e.u = make_float2((float)x, (float)y);
        e.s = make_float2((float)lg[px], (float)lg[HW + px]);
        for (int d = 0; d < 3; ++d) e.X[d] = xyz ? (float)xyz[d * HW + px] : 0.f;  // null: the selection alone (the points are decoded for the selected rows afterwards)
        v = vis ? (float)vis[px] : 0.f;
        e.g = 0;
        e.src = n;
        return e;
    }
    __device__ __forceinline__ Entry finish(Entry e, float v) const {
        e.s = make_float2(__expf(e.s.x - lse) * scale, __expf(e.s.y - lse) * scale);
        for (int d = 0; d < 3; ++d) e.X[d] *= ns[d];
        e.g = vis ? ((1.f / (1.f + expf(-v))) > vis_thresh ? 1 : 0) : 0;  // torch.sigmoid's own formula
        return e;
    }
    __device__ __forceinline__ Entry load(int n) const {
        float v;
        const Entry e = fetch(n, v);
        return finish(e, v);
    }
};

// torch.quantile's threshold of the row whose n order-preserving integer keys the calling workgroup (kThreads threads) has just written to
// `keys` (LDS; no barrier yet); segc: this thread's count of visible entries (mode 2).  Same value in every thread.
// torch.quantile needs only the two order statistics around q (n - 1): a most-significant-digit RADIX SELECT over the order-preserving
// integer image of the weights (4 passes of 8 bits) finds the lower one in O(n), one counting pass the upper one -- the bitonic sort it
// replaces took 78 barrier-separated stages for n = 4096.  The threshold is formed from the same two floats, so the selected index sets
// are unchanged bit for bit.  Barriers are what this costs (16 wavefronts each): every pass has ONE -- the histogram of pass k is scanned
// by every wavefront for itself (no broadcast), the histogram of pass k+1 was zeroed while pass k was counting, and with three of them in
// rotation the one zeroed during pass k+2 is the one whose scan ended before pass k+1's barrier.
__device__ __forceinline__ unsigned weight_key(float w) {
    const unsigned u = __float_as_uint(w);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending unsigned order == ascending float order
}
__device__ __forceinline__ float key_weight(unsigned key) { return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key); }  // its inverse
// key_at(j, i): key of entry i = tid + j * kThreads (from LDS, or from the caller's registers: then ITERS = the number of entries a thread
// holds, and the walks over them are unrolled so that j is a compile-time register index)
template <int ITERS = 0, class KeyAt>
__device__ __forceinline__ float quantile_threshold(const SelectParams& p, int n, KeyAt&& key_at, int segc) {
    __shared__ int wave_seg[kWaves];    // mode 2: visible entries per wavefront
    __shared__ int wave_le[kWaves];     // upper order statistic: keys <= the lower one / smallest key above it, per wavefront
    __shared__ unsigned wave_above[kWaves];
    __shared__ int hist[3][256];        // radix passes rotate through three histograms: ONE barrier per pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 256) hist[0][tid] = 0;
    if (p.mode == 2) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) segc += __shfl_xor(segc, m, kWave);
        if (lane == 0) wave_seg[wave] = segc;
    }
    __syncthreads();
    float q = p.quantile;
    if (p.mode == 2) {
        int seg_total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) seg_total += wave_seg[w];
        q = 1.f - p.one_minus_q * ((float)seg_total / (float)n);  // test.py:102-103, fp32 like the tensor op
    }
    q = fminf(fmaxf(q, 0.f), 1.f);
    const float rank = q * (float)(n - 1);
    const float lo = floorf(rank), hi = ceilf(rank);
    const int klo = (int)lo, khi = min((int)hi, n - 1);
    // rank klo by the radix select; rank khi = klo + 1 is then either the same key (a duplicate: more than klo + 1 keys are <= it)
    // or the smallest key above it -- one counting / min pass instead of four more histogram passes
    unsigned found[2] = {0u, 0u};
    {
        unsigned prefix = 0u, mask = 0u;
        int k = klo;  // 0-based rank among the elements that still match the prefix
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = 8 * pass, cur = (3 - pass) % 3, nxt = (cur + 1) % 3;
            if (tid < 256) hist[nxt][tid] = 0;
            if constexpr (ITERS > 0) {
#pragma unroll
                for (int j = 0; j < ITERS; ++j) {
                    const int i = j * kThreads + tid;
                    if (j * kThreads >= n) break;  // uniform
                    const unsigned key = i < n ? key_at(j, i) : 0u;
                    hist_add(hist[cur], (key >> shift) & 255u, i < n && (key & mask) == prefix, lane, pass == 3);
                }
            } else {
                for (int i0 = 0, j = 0; i0 < n; i0 += kThreads, ++j) {
                    const int i = i0 + tid;
                    const unsigned key = i < n ? key_at(j, i) : 0u;
                    hist_add(hist[cur], (key >> shift) & 255u, i < n && (key & mask) == prefix, lane, pass == 3);
                }
            }
            __syncthreads();
            // every wavefront: lane l owns bins 4l .. 4l+3, exclusive prefix over the lanes, then the bin holding rank k
            const int c0 = hist[cur][4 * lane], c1 = hist[cur][4 * lane + 1], c2 = hist[cur][4 * lane + 2], c3 = hist[cur][4 * lane + 3];
            const int mine = c0 + c1 + c2 + c3;
            int incl = mine;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int up = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += up;
            }
            const int excl = incl - mine;
            const bool hit = k >= excl && k < incl;  // exactly one lane
            int r = k - excl, bin = 4 * lane;
            if (r >= c0) { r -= c0; ++bin; if (r >= c1) { r -= c1; ++bin; if (r >= c2) { r -= c2; ++bin; } } }
            const int owner = __ffsll((long long)__ballot(hit)) - 1;
            prefix |= (unsigned)__shfl(bin, owner, kWave) << shift;
            mask |= 255u << shift;
            k = __shfl(r, owner, kWave);
        }
        found[0] = found[1] = prefix;
    }
    if (khi != klo) {  // uniform
        int le = 0;
        unsigned above = 0xFFFFFFFFu;
        auto count = [&](int j, int i) {
            const unsigned key = key_at(j, i);
            if (key <= found[0]) ++le;
            else above = min(above, key);
        };
        if constexpr (ITERS > 0) {
#pragma unroll
            for (int j = 0; j < ITERS; ++j)
                if (j * kThreads + tid < n) count(j, j * kThreads + tid);
        } else {
            for (int i = tid, j = 0; i < n; i += kThreads, ++j) count(j, i);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            le += __shfl_xor(le, m, kWave);
            above = min(above, (unsigned)__shfl_xor((int)above, m, kWave));
        }
        if (lane == 0) { wave_le[wave] = le; wave_above[wave] = above; }
        __syncthreads();
        int le_total = 0;
        unsigned above_min = 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { le_total += wave_le[w]; above_min = min(above_min, wave_above[w]); }
        if (le_total <= khi) found[1] = above_min;  // no duplicate reaches rank khi: the next distinct key
    }
    const float vlo = key_weight(found[0]), vhi = khi != klo ? key_weight(found[1]) : vlo;
    return torch_lerp(vlo, vhi, rank - lo);  // every thread forms it from the same two floats
}

// Order-preserving compaction of a row, 1024 candidates (one per thread) at a time: slot(keep, more) returns where this thread's entry
// goes (wave ballots + prefix over the wavefronts); `running` = survivors so far, the same value in every thread.  more: another call
// follows (its barrier keeps the per-wave counts of this one alive until everybody has read them).
struct Compactor {
    int running = 0;
    __device__ __forceinline__ int slot(bool keep, bool more) {
        __shared__ int wave_cnt[kWaves];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = running, tot = running;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            if (w < wave) off += wave_cnt[w];
            tot += wave_cnt[w];
        }
        running = tot;
        if (more) __syncthreads();  // wave_cnt is rewritten by the next call
        return off + before;
    }
};

// The selection of row b by the calling workgroup (kThreads threads).  ec: entries tid + k * 1024 of the row, already in registers -- all
// this thread handles when N <= 4096: their weights feed the threshold search, and their coordinates are there by the time the
// compaction knows where they go.  srt: n floats of LDS (modes 1, 2).
constexpr int kCache = 4;  // entries tid, tid + 1024, ... a thread keeps in registers (rows of up to 4096 candidates never re-read one)

template <class Source>
__device__ __forceinline__ void select_row(const SelectParams& p, int b, int n, const Source& src, const Entry (&ec)[kCache], float* srt) {
    const int tid = threadIdx.x;
    const size_t base = (size_t)b * p.N;
    auto weight_of = [&](const Entry& e) { return (p.mode == 2 && !e.g) ? 0.f : e.s.x + e.s.y; };  // mode 2: (inv_std * seg).sum(-1)
    float thr = -FLT_MAX;
    if (p.mode != 0 && n > 0) {
        unsigned* keys = reinterpret_cast<unsigned*>(srt);
        int segc = 0;
        auto stage = [&](const Entry& e, int i) {
            keys[i] = weight_key(weight_of(e));
            if (p.mode == 2 && e.g) ++segc;
        };
#pragma unroll
        for (int k = 0; k < kCache; ++k)
            if (tid + k * kThreads < n) stage(ec[k], tid + k * kThreads);
        for (int i = tid + kCache * kThreads; i < n; i += kThreads) stage(src.load(i), i);
        thr = quantile_threshold(p, n, [&](int, int i) { return keys[i]; }, segc);
    }
    const RowCopy rows{nullptr, nullptr, nullptr, nullptr, p.o_pts2d, p.o_w, p.o_pts3d, p.o_index, p.square};
    auto put = [&](const Entry& e, int o) { rows.entry_from(base, o, e.u.x, e.u.y, e.s, e.X[0], e.X[1], e.X[2], e.src); };
    Compactor cmp;
    auto chunk = [&](int i0, const Entry& e, bool mine) {  // mine: this thread has an entry in the chunk (e)
        bool keep = false;
        if (mine) {
            if (p.mode == 0) keep = e.g != 0;
            else keep = weight_of(e) >= thr && (p.mode == 1 || e.g != 0);
        }
        const int o = cmp.slot(keep, i0 + kThreads < n);
        if (keep) put(e, o);
    };
#pragma unroll
    for (int k = 0; k < kCache; ++k) {
        if (k * kThreads >= n) break;  // uniform
        chunk(k * kThreads, ec[k], tid + k * kThreads < n);
    }
    for (int i0 = kCache * kThreads; i0 < n; i0 += kThreads) {
        const bool mine = i0 + tid < n;
        chunk(i0, mine ? src.load(i0 + tid) : ec[0], mine);
    }
    const int total = pad_rows(p.pose0 + b, n, cmp.running, p.min_count, p.seed, [&](int i, int k) { put(src.load(i), k); });  // test.py:108-113
    if (tid == 0) p.counts[b] = total;
}

__global__ __launch_bounds__(kThreads) void lc_dense_select_kernel(const SelectParams p) {
    extern __shared__ float srt[];  // n order-preserving integer keys of the weights (modes 1, 2)
    const int b = blockIdx.x, tid = threadIdx.x;
    const ArraySource src{p, (size_t)b * p.N};
    // requested whole before the row's count is known: one memory round trip at the start instead of three dependent ones
    Entry ec[kCache] = {};
#pragma unroll
    for (int k = 0; k < kCache; ++k)
        if (tid + k * kThreads < p.N) ec[k] = src.load(tid + k * kThreads);
    const int n = min(p.in_counts ? p.in_counts[b] : p.N, p.N);
    select_row(p, b, n, src, ec, srt);
}

// Test time, N <= 16384 sampled pixels per object: the dense front end (joint softmax x scale, strided sub-sampling, visibility mask:
// test.py:85-92) and the point selection (test.py:94-113) in ONE launch, one workgroup per object.  The front end's (B,N,.) arrays
// are never written: every thread forms the entry of its own sampled pixel in registers (beyond 1024 pixels per object: the further
// ones again from the maps where the selection needs them) -- the log-sum-exp by the front end's own 512
// threads in the front end's own order, so every selected value equals what the two launches produce bit for bit -- and hands it to
// the selection above.
// Rows of more than kCache * 1024 candidates per object (zlmo's test-time shape: 128x128 at stride 1 = 16 per thread).  Walking the entries
// beyond the cached four one at a time -- form an entry from the maps, use it, form the next -- was a dependent memory round trip per entry
// and phase: 12 in the staging loop, 12 in the compaction, 55 us per 64 objects.  Here a thread requests what a phase needs of ALL its entries
// at once, branch-free (an index behind the row re-requests the row's last entry, a cache hit): the two weight logits and the visibility
// logit of its 16 pixels before the log-sum-exp, and -- once the threshold is known -- the xyz of eight entries at a time.  Weights and
// visibility bits stay in registers in between (32 + 1); every value is formed exactly as MapSource forms it: same results bit for bit.
constexpr int kWideCache = 16, kWideBatch = 8;
template <typename T, typename TX>
__device__ __forceinline__ void select_row_wide(const SelectParams& p, int b, MapSource<T, TX>& src, const int two_hw, float (*red)[2], float* srt) {
    const int tid = threadIdx.x, n = p.N;
    const size_t base = (size_t)b * p.N;
    LC_FS_STAMP(0);
    float2 w[kWideCache];   // raw logits, then the weights
    float vraw[kWideCache];
    unsigned gbits = 0u;
#pragma unroll
    for (int k = 0; k < kWideCache; ++k) {
        int x, y;
        const int px = src.pixel(min(tid + k * kThreads, n - 1), x, y);
        w[k] = make_float2((float)src.lg[px], (float)src.lg[src.HW + px]);
        vraw[k] = src.vis ? (float)src.vis[px] : 0.f;
    }
    src.lse = block_lse<kDenseLseThreads>(src.lg, two_hw, red);
    LC_FS_STAMP(1);
#pragma unroll
    for (int k = 0; k < kWideCache; ++k) {
        w[k] = make_float2(__expf(w[k].x - src.lse) * src.scale, __expf(w[k].y - src.lse) * src.scale);
        if (src.vis && (1.f / (1.f + expf(-vraw[k]))) > src.vis_thresh) gbits |= 1u << k;  // torch.sigmoid's own formula
    }
    auto weight_of = [&](int k) { return (p.mode == 2 && !((gbits >> k) & 1u)) ? 0.f : w[k].x + w[k].y; };
    float thr = -FLT_MAX;
    if (p.mode != 0) {
        // the keys never leave the registers: a thread's entry j of every pass is its own w[j] (the generic path stages them in LDS)
        const int mine = min(kWideCache, max(0, (n - tid + kThreads - 1) / kThreads));  // entries tid, tid + 1024, ... below n
        const int segc = p.mode == 2 ? __popc(gbits & ((1u << mine) - 1u)) : 0;        // (a slot behind the row repeats the row's last entry)
        LC_FS_STAMP(2);
#ifdef LC_SELECT_SKIP_SEARCH  // timing experiment only (wrong results; scripts/ubench/select_skip_ab.py): the kernel without the threshold search
        thr = 1e-4f * (float)segc * 0.f + 3e-5f;
#else
        thr = quantile_threshold<kWideCache>(p, n, [&](int j, int) { return weight_key(weight_of(j)); }, segc);  // j: a compile-time index there
#endif
    }
    LC_FS_STAMP(3);
    const RowCopy rows{nullptr, nullptr, nullptr, nullptr, p.o_pts2d, p.o_w, p.o_pts3d, p.o_index, p.square};
    // Order-preserving compaction of all 16 chunks of 1024 candidates with THREE barriers (one barrier pair per chunk was 32 of them,
    // 16 wavefronts each): the keep flags of a thread's 16 entries are known at once, so every wavefront posts its 16 per-chunk counts,
    // the first 256 threads scan the 16 x 16 table in (chunk, wavefront) order, and a thread's entry k goes to
    // offs[k][wave] + (kept entries of chunk k in lower lanes of its wavefront).
    __shared__ int cnt[kWideCache][kWaves], offs[kWideCache][kWaves], wtot[4];
    const int lane = tid & 63, wave = tid >> 6;
    unsigned keepbits = 0u;
    unsigned before[kWideCache / 4] = {};  // 8 bits per chunk
#pragma unroll
    for (int k = 0; k < kWideCache; ++k) {
        const int i = tid + k * kThreads;
        bool keep = false;
        if (i < n) {
            const bool g = (gbits >> k) & 1u;
            keep = p.mode == 0 ? g : (weight_of(k) >= thr && (p.mode == 1 || g));
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) keepbits |= 1u << k;
        before[k >> 2] |= (unsigned)__popcll(bal & ((1ull << lane) - 1ull)) << (8 * (k & 3));
        if (lane == 0) cnt[k][wave] = __popcll(bal);
    }
    __syncthreads();
    int v = 0, incl = 0;
    if (tid < kWideCache * kWaves) {
        v = cnt[tid / kWaves][tid % kWaves];
        incl = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int up = __shfl_up(incl, d, kWave);
            if (lane >= d) incl += up;
        }
        if (lane == kWave - 1) wtot[wave] = incl;
    }
    __syncthreads();
    static_assert(kWideCache * kWaves == 4 * kWave, "the table is scanned by four wavefronts");
    if (tid < kWideCache * kWaves) {
        int pre = 0;
        for (int w = 0; w < wave; ++w) pre += wtot[w];
        offs[tid / kWaves][tid % kWaves] = pre + incl - v;
    }
    const int kept = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    LC_FS_STAMP(4);
#pragma unroll
    for (int h = 0; h < kWideCache; h += kWideBatch) {
        if (h * kThreads >= n) break;  // uniform
        float X[kWideBatch][3];
#pragma unroll
        for (int j = 0; j < kWideBatch; ++j) {
            int x, y;
            const int px = src.pixel(min(tid + (h + j) * kThreads, n - 1), x, y);
#pragma unroll
            for (int d = 0; d < 3; ++d) X[j][d] = src.xyz ? (float)src.xyz[d * src.HW + px] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < kWideBatch; ++j) {
            const int k = h + j, i = k * kThreads + tid;
            if ((keepbits >> k) & 1u) {
                int x, y;
                src.pixel(i, x, y);  // (recomputed for the survivors rather than held for everybody: registers)
                const int o = offs[k][wave] + (int)((before[k >> 2] >> (8 * (k & 3))) & 0xffu);
                rows.entry_from(base, o, (float)x, (float)y, w[k], X[j][0] * src.ns[0], X[j][1] * src.ns[1], X[j][2] * src.ns[2], i);
            }
        }
    }
    const int total = pad_rows(p.pose0 + b, n, kept, p.min_count, p.seed, [&](int i, int k) {
        const Entry e = src.load(i);
        rows.entry_from(base, k, e.u.x, e.u.y, e.s, e.X[0], e.X[1], e.X[2], e.src);
    });  // test.py:108-113
    if (tid == 0) p.counts[b] = total;
    LC_FS_STAMP(5);
}

constexpr int kSelRowWords = 256;  // the split form's exchange rows (below)
constexpr size_t kSelSplitPoseBytes = 2 * kSplitMaxParts * kSelRowWords * sizeof(unsigned long long) + 128;

template <typename T, typename TX, bool WIDE>
__global__ __launch_bounds__(kThreads) void lc_dense_frontend_select_kernel(const SelectParams p, const DenseParams d) {
    extern __shared__ float srt[];
    __shared__ float red[kDenseLseThreads / kWave][2];
    const int b = blockIdx.x, tid = threadIdx.x, HW = d.H * d.W;
    MapSource<T, TX> src;
    src.lg = static_cast<const T*>(d.wlogits) + (size_t)b * d.wl_bs;
    src.xyz = d.xyz ? static_cast<const TX*>(d.xyz) + (size_t)b * d.xyz_bs : nullptr;
    src.vis = d.vis_logits ? static_cast<const T*>(d.vis_logits) + (size_t)b * d.vis_bs : nullptr;
    src.vis_thresh = d.vis_thresh;
    src.HW = HW; src.W = d.W; src.top = d.top; src.left = d.left; src.sample = d.sample;
    src.Wn = (d.W - d.left + d.sample - 1) / d.sample;
    src.lse = 0.f;
    src.scale = map_scalar_at(d.wscale, d.wscale_dtype, b);
    for (int k = 0; k < 3; ++k) src.ns[k] = d.noc_scale ? d.noc_scale[3 * b + k] : 1.f;
    if (p.split_ws) {  // (uniform) the rescue launch behind a split launch: only the objects a part gave up on
        char* region = static_cast<char*>(p.split_ws) + (size_t)b * kSelSplitPoseBytes;
        const bool marked = xcd_load(split_tail(region, kSelSplitPoseBytes) + 1) != 0u;
        if (!split_rescue_enter(region, kSelSplitPoseBytes, marked, tid, kThreads)) return;
    }
    if constexpr (WIDE) {
        select_row_wide(p, b, src, 2 * HW, red, srt);
    } else {
        Entry ec[kCache] = {};
        float vraw[kCache] = {};
#pragma unroll
        for (int k = 0; k < kCache; ++k)
            if (tid + k * kThreads < p.N) ec[k] = src.fetch(tid + k * kThreads, vraw[k]);  // in flight while the log-sum-exp is formed
        src.lse = block_lse<kDenseLseThreads>(src.lg, 2 * HW, red);
#pragma unroll
        for (int k = 0; k < kCache; ++k) ec[k] = src.finish(ec[k], vraw[k]);
        select_row(p, b, p.N, src, ec, srt);
    }
}

// ---- Front end + selection of ONE object by several workgroups (rows of more than 4096 candidates, at most 128 objects) -----------------------
// One workgroup per object pulls the object's maps (128x128: 196 KB of logits) through ONE compute unit and walks 16 candidates per thread
// through every phase: 36.7 us at zlmo's test-time shape, 24 of them without the threshold search, on a quarter of the chip.  Here part g of
// P = 2 / 4 / 8 owns the candidates [4096 g, 4096 (g + 1)) -- four per thread, in registers, the shape the narrow kernel above was built for --
// and the parts meet five to seven times through ticketed words in the workspace (lc_common.h: SplitSum; rows of 256 words here):
//   1. log-sum-exp: part g plays threads [512 g / P, 512 (g + 1) / P) of the front end's 512-thread reduction (lse_thread_share: same groups,
//      same order), the 8 wavefront pairs meet and are merged in wavefront order -- the SAME float as block_lse; the visible counts ride along;
//   2. radix select: every pass counts its own keys, the 256 bins meet (one word per thread), every part scans the merged histogram;
//      the upper order statistic is read off the last pass' merged bins (exact keys) -- or, the rank closing its prefix, one more meeting;
//   3. compaction: the parts' survivor counts meet, part g writes behind the parts before it.
// Every value is formed by the expressions of the one-workgroup kernels and every decision is taken on the same integers: outputs bit for bit
// those of lc_dense_frontend_select_kernel (tests/test_gpu_select.py).  The parts wait for each other, for a bounded time: a part that has
// waited in vain marks the object's region (lc_common.h: SplitSum's tail) and leaves; the launch is always followed by the one-workgroup
// kernel in its rescue role (its SelectParams keep split_ws), whose workgroups leave at once unless their object is marked -- then they zero its region
// and select it themselves, bit for bit what the parts would have written.  The grid is sized to one workgroup per compute unit so that the
// parts normally do meet; anything else holding compute units costs time, not objects.

// The parts of an object meet: threads tid < nw publish word tid of this part (`mine`); the nw x G words of all parts are then fetched by the
// workgroup's threads one word each (thread t: word t % nw of part t / nw -- one or two registers per thread, not G) and handed to
// consume(g, l, word).  With every thread of a few polling G words each the kernel took 28.3 us, spread like this 25.6.
// false: a part never arrived (bounded wait, lc_common.h).
template <class F>
__device__ __forceinline__ bool parts_meet(SplitSum& sx, int tid, int nw, unsigned mine, F&& consume) {
    const unsigned long long ticket = (unsigned long long)(sx.base + sx.seq + 1u) << 32;
    unsigned long long* rows = sx.exch + (size_t)(sx.seq & 1u) * (kSplitMaxParts * kSelRowWords);
    ++sx.seq;
    if (tid < nw) xcd_store(rows + sx.part * kSelRowWords + tid, ticket | mine);
    bool ok = true;
    for (int idx = tid; idx < nw * sx.G; idx += kThreads) {
        const int g = idx / nw, l = idx - g * nw;
        const unsigned long long* at = rows + g * kSelRowWords + l;
        unsigned long long word = 0, t0 = 0;
        for (int polls = 0;; ++polls) {  // (several reads in flight per word, a sleep apart, were measured slower: 25.6 us with one, 26.1 / 27.1 / 27.8 with 2 / 3 / 5)
            word = xcd_load(at);
            if (((word ^ ticket) >> 32) == 0) break;
            if (split_wait_expired(polls, t0)) { ok = false; break; }  // the same wall-clock budget as the split solve's wait (lc_common.h)
        }
        if (ok) consume(g, l, (unsigned)word);
    }
    return ok;
}

template <typename T, typename TX>
__global__ __launch_bounds__(kThreads) void lc_dense_frontend_select_split_kernel(const SelectParams p, const DenseParams d) {
    constexpr int kLseWaves = kDenseLseThreads / kWave;
    __shared__ float red[kLseWaves][2];
    __shared__ int wave_seg[kWaves], seg_parts[kSplitMaxParts], kept_parts[kSplitMaxParts], never_arrived;
    __shared__ unsigned wave_above[kWaves], above_parts[kSplitMaxParts];
    __shared__ int hist[3][256], merged[3][256];
    __shared__ int cnt[kCache][kWaves];
    const int P = p.split_parts, b = blockIdx.x / P, part = blockIdx.x % P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, HW = d.H * d.W, n = p.N;
    const size_t base = (size_t)b * p.N;
    MapSource<T, TX> src;
    src.lg = static_cast<const T*>(d.wlogits) + (size_t)b * d.wl_bs;
    src.xyz = d.xyz ? static_cast<const TX*>(d.xyz) + (size_t)b * d.xyz_bs : nullptr;
    src.vis = d.vis_logits ? static_cast<const T*>(d.vis_logits) + (size_t)b * d.vis_bs : nullptr;
    src.vis_thresh = d.vis_thresh;
    src.HW = HW; src.W = d.W; src.top = d.top; src.left = d.left; src.sample = d.sample;
    src.Wn = (d.W - d.left + d.sample - 1) / d.sample;
    src.lse = 0.f;
    src.scale = map_scalar_at(d.wscale, d.wscale_dtype, b);
    for (int k = 0; k < 3; ++k) src.ns[k] = d.noc_scale ? d.noc_scale[3 * b + k] : 1.f;
    char* region = static_cast<char*>(p.split_ws) + (size_t)b * kSelSplitPoseBytes;
    unsigned* epoch = reinterpret_cast<unsigned*>(region + kSelSplitPoseBytes - 128);
    SplitSum sx{reinterpret_cast<unsigned long long*>(region), epoch, P, part, xcd_load(epoch), 0u, false};
    LC_FS_STAMP(0);
    if (tid == 0) never_arrived = 0;
    if (tid < 256) { hist[0][tid] = 0; merged[0][tid] = 0; }  // the first pass' pair (the later ones are zeroed a pass ahead)
    // this part's candidates: entries i0 + tid + 1024 k, requested while the log-sum-exp is formed
    const int i0 = part * kCache * kThreads;
    Entry ec[kCache] = {};
    float vraw[kCache] = {};
    bool have[kCache];
#pragma unroll
    for (int k = 0; k < kCache; ++k) {
        have[k] = i0 + tid + k * kThreads < n;
        if (have[k]) ec[k] = src.fetch(i0 + tid + k * kThreads, vraw[k]);
    }
    // 1. this part's threads of the front end's reduction
    const int per = kDenseLseThreads / P, wpp = per / kWave;  // threads / wavefronts of that reduction this part plays
    if (tid < per) {
        float m = -FLT_MAX, s = 0.f;
        lse_thread_share<kDenseLseThreads, 8>(src.lg, 2 * HW, part * per + tid, m, s);
        lse_wave_merge(m, s);
        if (lane == 0) { red[wave][0] = m; red[wave][1] = s; }
    }
    int segc = 0;
    if (p.mode == 2) {
#pragma unroll
        for (int k = 0; k < kCache; ++k)
            if (have[k] && src.vis && (1.f / (1.f + expf(-vraw[k]))) > src.vis_thresh) ++segc;  // MapSource::finish's own test
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) segc += __shfl_xor(segc, m, kWave);
        if (lane == 0) wave_seg[wave] = segc;
    }
    __syncthreads();
    auto meet = [&](int nw, unsigned mine, auto&& consume) {  // + the barrier that makes what `consume` wrote to LDS visible; false: give up
        if (!parts_meet(sx, tid, nw, mine, consume)) never_arrived = 1;
        __syncthreads();
        return never_arrived == 0;
    };
    auto give_up = [&]() {  // (uniform) this part waited in vain for another: the object is marked, the rescue launch behind this one selects it again
        if (tid == 0) xcd_store(sx.epoch + 1, 1u);
    };
    {
        unsigned mine = 0u;
        if (tid < 2 * wpp) mine = __float_as_uint(red[tid >> 1][tid & 1]);
        else if (tid == 2 * wpp) {
            int mine_seg = 0;
            if (p.mode == 2)
                for (int w = 0; w < kWaves; ++w) mine_seg += wave_seg[w];
            mine = (unsigned)mine_seg;
        }
        __syncthreads();  // red[] is rewritten with all eight pairs
        LC_FS_STAMP(1);
        if (!meet(2 * wpp + 1, mine, [&](int g, int l, unsigned word) {
                if (l < 2 * wpp) red[g * wpp + (l >> 1)][l & 1] = __uint_as_float(word);
                else seg_parts[g] = (int)word;
            })) {
            give_up();
            return;
        }
    }
    {
        float m = red[0][0], s = red[0][1];
#pragma unroll
        for (int w = 1; w < kLseWaves; ++w) ms_merge(m, s, red[w][0], red[w][1]);
        src.lse = m + __logf(s);  // block_lse's own last lines
    }
    LC_FS_STAMP(2);
    int seg_total = 0;
    for (int g = 0; g < P; ++g) seg_total += seg_parts[g];
#pragma unroll
    for (int k = 0; k < kCache; ++k) ec[k] = src.finish(ec[k], vraw[k]);
    auto weight_of = [&](const Entry& e) { return (p.mode == 2 && !e.g) ? 0.f : e.s.x + e.s.y; };
    // 2. the threshold (quantile_threshold above, the bins of every pass merged over the parts)
    float thr = -FLT_MAX;
    if (p.mode != 0) {
        unsigned key[kCache];
#pragma unroll
        for (int k = 0; k < kCache; ++k) key[k] = have[k] ? weight_key(weight_of(ec[k])) : 0u;
        float q = p.quantile;
        if (p.mode == 2) q = 1.f - p.one_minus_q * ((float)seg_total / (float)n);
        q = fminf(fmaxf(q, 0.f), 1.f);
        const float rank = q * (float)(n - 1);
        const float lo = floorf(rank), hi = ceilf(rank);
        const int klo = (int)lo, khi = min((int)hi, n - 1);
        LC_FS_STAMP(3);
        unsigned prefix = 0u, mask = 0u, found[2];
        int k = klo, in_bin = 0, next_bin = 256;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = 8 * pass, cur = (3 - pass) % 3, nxt = (cur + 1) % 3;
            const int* all = merged[cur];  // the parts' bins added up (three in rotation, zeroed a pass ahead, like the histograms)
            if (tid < 256) { hist[nxt][tid] = 0; merged[nxt][tid] = 0; }
#pragma unroll
            for (int j = 0; j < kCache; ++j) hist_add(hist[cur], (key[j] >> shift) & 255u, have[j] && (key[j] & mask) == prefix, lane, pass == 3);
            __syncthreads();
            if (!meet(256, tid < 256 ? (unsigned)hist[cur][tid] : 0u, [&](int, int l, unsigned word) { if (word) atomicAdd(&merged[cur][l], (int)word); })) {
                give_up();
                return;
            }
            // every wavefront: lane l owns bins 4l .. 4l+3, exclusive prefix over the lanes, then the bin holding rank k
            const int c0 = all[4 * lane], c1 = all[4 * lane + 1], c2 = all[4 * lane + 2], c3 = all[4 * lane + 3];
            const int mine4 = c0 + c1 + c2 + c3;
            int incl = mine4;
#pragma unroll
            for (int dd = 1; dd < kWave; dd <<= 1) {
                const int up = __shfl_up(incl, dd, kWave);
                if (lane >= dd) incl += up;
            }
            const int excl = incl - mine4;
            const bool hit = k >= excl && k < incl;  // exactly one lane
            int r = k - excl, bin = 4 * lane, cb = c0;
            if (r >= c0) { r -= c0; ++bin; cb = c1; if (r >= c1) { r -= c1; ++bin; cb = c2; if (r >= c2) { r -= c2; ++bin; cb = c3; } } }
            const int owner = __ffsll((long long)__ballot(hit)) - 1;
            bin = __shfl(bin, owner, kWave);
            prefix |= (unsigned)bin << shift;
            mask |= 255u << shift;
            k = __shfl(r, owner, kWave);
            if (pass == 0) {  // the bins are exact keys: how many copies of the found key, and the next key present under the same 24-bit prefix
                in_bin = __shfl(cb, owner, kWave);
                int nb = 256;
                if (4 * lane + 3 > bin) {
                    if (c3 > 0 && 4 * lane + 3 > bin) nb = 4 * lane + 3;
                    if (c2 > 0 && 4 * lane + 2 > bin) nb = 4 * lane + 2;
                    if (c1 > 0 && 4 * lane + 1 > bin) nb = 4 * lane + 1;
                    if (c0 > 0 && 4 * lane > bin) nb = 4 * lane;
                }
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) nb = min(nb, __shfl_xor(nb, mm, kWave));
                next_bin = nb;
            }
        }
        found[0] = found[1] = prefix;
        if (khi != klo && k + 1 >= in_bin) {  // (uniform) rank khi = klo + 1 is the next distinct key
            if (next_bin < 256) {
                found[1] = (prefix & ~255u) | (unsigned)next_bin;
            } else {  // none under this prefix: the smallest key above, over all parts
                unsigned above = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < kCache; ++j)
                    if (have[j] && key[j] > found[0]) above = min(above, key[j]);
#pragma unroll
                for (int mm = 32; mm >= 1; mm >>= 1) above = min(above, (unsigned)__shfl_xor((int)above, mm, kWave));
                if (lane == 0) wave_above[wave] = above;
                __syncthreads();
                unsigned mine = 0xFFFFFFFFu;
                for (int w = 0; w < kWaves; ++w) mine = min(mine, wave_above[w]);
                if (!meet(1, mine, [&](int g, int, unsigned word) { above_parts[g] = word; })) {
                    give_up();
                    return;
                }
                unsigned above_min = 0xFFFFFFFFu;
                for (int g = 0; g < P; ++g) above_min = min(above_min, above_parts[g]);
                found[1] = above_min;
            }
        }
        const float vlo = key_weight(found[0]), vhi = khi != klo ? key_weight(found[1]) : vlo;
        thr = torch_lerp(vlo, vhi, rank - lo);
    }
    LC_FS_STAMP(4);
    // 3. compaction: this part's survivors behind those of the parts before it
    bool keep[kCache];
    unsigned long long bal[kCache];
#pragma unroll
    for (int k = 0; k < kCache; ++k) {
        keep[k] = have[k] && (p.mode == 0 ? ec[k].g != 0 : (weight_of(ec[k]) >= thr && (p.mode == 1 || ec[k].g != 0)));
        bal[k] = __ballot(keep[k]);
        if (lane == 0) cnt[k][wave] = __popcll(bal[k]);
    }
    __syncthreads();
    int off[kCache], mine_total = 0;
#pragma unroll
    for (int k = 0; k < kCache; ++k) {
        off[k] = mine_total;
        for (int w = 0; w < kWaves; ++w) {
            if (w < wave) off[k] += cnt[k][w];
            mine_total += cnt[k][w];
        }
    }
    if (!meet(1, (unsigned)mine_total, [&](int g, int, unsigned word) { kept_parts[g] = (int)word; })) {
        give_up();
        return;
    }
    LC_FS_STAMP(5);
    int before = 0, kept = 0;
    for (int g = 0; g < P; ++g) {
        if (g < part) before += kept_parts[g];
        kept += kept_parts[g];
    }
    const RowCopy rows{nullptr, nullptr, nullptr, nullptr, p.o_pts2d, p.o_w, p.o_pts3d, p.o_index, p.square};
    auto put = [&](const Entry& e, int o) { rows.entry_from(base, o, e.u.x, e.u.y, e.s, e.X[0], e.X[1], e.X[2], e.src); };
#pragma unroll
    for (int k = 0; k < kCache; ++k)
        if (keep[k]) put(ec[k], before + off[k] + __popcll(bal[k] & ((1ull << lane) - 1ull)));
    if (part != 0) return;
    const int total = pad_rows(p.pose0 + b, n, kept, p.min_count, p.seed, [&](int i, int k) { put(src.load(i), k); });  // test.py:108-113
    LC_FS_STAMP(6);
    if (tid == 0) {
        p.counts[b] = total;
        xcd_store(sx.epoch, sx.base + sx.seq);  // the tickets of this object's next launch count on from here
    }
}

}  // namespace

int dense_select_split_parts(int B, int N) {
    if (N <= kCache * kThreads || N > kFusedSelectMaxPoints || B <= 0) return 1;
    const int P = split_parts_for(B);           // at most one workgroup per compute unit: all resident together
    return N <= P * kCache * kThreads ? P : 1;  // a part's candidates are the four per thread its registers hold
}
size_t dense_select_split_workspace_bytes(int B, int N) { return dense_select_split_parts(B, N) > 1 ? (size_t)B * kSelSplitPoseBytes : 0; }

int launch_dense_select(const SelectParams& p, hipStream_t stream) {
    if (p.B <= 0) return 0;
    int P = 1;
    while (P < p.N) P <<= 1;
    const size_t lds = p.mode == 0 ? 0 : (size_t)P * sizeof(float);
    if (lds > 128 * 1024) return 3;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(lc_dense_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 2;
    hipLaunchKernelGGL(lc_dense_select_kernel, dim3(p.B), dim3(kThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int launch_dense_frontend_select(const SelectParams& p, const DenseParams& d_in, hipStream_t stream) {
    if (p.B <= 0) return 0;
    const DenseParams d = with_dense_strides(d_in);
    if (p.N > kFusedSelectMaxPoints) return 3;
    int P = kThreads;
    while (P < p.N) P <<= 1;
    const size_t lds = p.mode == 0 ? 0 : (size_t)P * sizeof(float);
    if (d.xyz_dtype != d.map_dtype && d.xyz_dtype != kMapF32) return 2;
    int rc = 0;
    bool rescue = false;
    SelectParams pr;
    auto go = [&](auto* kernel) {
        if (lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            rc = 2;
        else hipLaunchKernelGGL(kernel, dim3(p.B), dim3(kThreads), lds, stream, pr, d);
    };
    if (p.split_ws) {
        const int parts = dense_select_split_parts(p.B, p.N);
        if (parts > 1) {  // several workgroups per object, four candidates per thread (keys in registers: no dynamic LDS)
            SelectParams ps = p;
            ps.split_parts = parts;
            auto split = [&](auto* kernel) { hipLaunchKernelGGL(kernel, dim3((unsigned)p.B * parts), dim3(kThreads), 0, stream, ps, d); };
            LC_MAP_DISPATCH(d.map_dtype, if (d.xyz_dtype == d.map_dtype) split(lc_dense_frontend_select_split_kernel<T, T>);
                                         else split(lc_dense_frontend_select_split_kernel<T, float>));
            if (hipGetLastError() != hipSuccess) return 2;
            rescue = true;  // ... and the one-workgroup kernel behind it, for the objects a part gave up on (p.split_ws stays set)
        }
    }
    pr = p;
    if (!rescue) pr.split_ws = nullptr;
    const bool wide = p.N > kCache * kThreads;  // more candidates per thread than the register cache of the one-entry-at-a-time walk holds
    LC_MAP_DISPATCH(d.map_dtype,
                    if (d.xyz_dtype == d.map_dtype) { if (wide) go(lc_dense_frontend_select_kernel<T, T, true>); else go(lc_dense_frontend_select_kernel<T, T, false>); }
                    else { if (wide) go(lc_dense_frontend_select_kernel<T, float, true>); else go(lc_dense_frontend_select_kernel<T, float, false>); });
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace lc

#ifdef LC_SELECT_STAMPS
extern "C" __attribute__((visibility("default"))) int lc_debug_select_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(lc::select_diag::g_stamp), sizeof(unsigned long long) * 10) == hipSuccess ? 0 : 1;
}
#endif
