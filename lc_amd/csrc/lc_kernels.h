// Internal launch interface between the C ABI (lc_capi.hip) and the kernels. Not installed; see include/lc_amd.h.
#pragma once
#include <hip/hip_runtime.h>

namespace lc {

constexpr int kLossAuxStride = 40;
// up to this many one-wave workgroups use the 'latency' build (one wave per SIMD: 296 registers under the max-ILP schedule, no
// spills; lc_pnp_latency.hip / lc_fused_latency.hip); beyond it -- more workgroups than SIMDs -- the two-waves-per-SIMD build wins
// (round 1: 2048 workgroups 57 us with the latency build vs 19.5 us with the other; round 3, threshold 768 -> 1024: 896 workgroups
// 16.4 -> 15.8 us, 1024 workgroups 18.2 -> 18.0 us)
#ifndef LC_BIG_WPS
#define LC_BIG_WPS 2  // waves per SIMD the large-grid builds of the pose kernels are register-limited to (A/B: scripts/ubench/pnp_ab.py)
#endif
#ifndef LC_LATENCY_GRID_MAX
#define LC_LATENCY_GRID_MAX 1024
#endif
constexpr int kLatencyGridMax = LC_LATENCY_GRID_MAX;

struct LossParams {
    const float* K;         // (B,3,3)
    const float* pose;      // (B,7) w,x,y,z,tx,ty,tz
    const float* pts3d;     // (B,N,3)
    const float* pts2d;     // (B,N,2)
    const float* inv_std;   // (B,N,2)
    const float* valid;     // (B,N) or null
    const float* bbox;      // (B,8,3)
    const float* grad_out;  // (B,) or null (== ones)
    float* loss;            // (B,)
    float* d_pts2d;         // (B,N,2) or null (forward only)
    float* d_inv_std;       // (B,N,2)
    float* d_pts3d;         // (B,N,3) or null
    float* aux;             // (B,kLossAuxStride) or null
    int B, N;
    float max_err_len, rel_thresh, w_e_thresh;
    int cov_2d;  // 0: 3D bbox-corner covariance (every reference call site); 1: projected corners (cov_mixed.py:125-127)
    void* workspace;         // tiled form (N > 256): cov_loss_workspace_bytes(B, N) bytes, zeroed once by the caller, or null
    size_t workspace_bytes;
};
int launch_cov_loss(const LossParams& p, hipStream_t stream);  // 3: workspace too small
size_t cov_loss_workspace_bytes(int B, int N);  // 0: the shape does not use the tiled form
// tiles, workgroups per sample, tiles per workgroup of the tiled form; reserved_groups: compute units to leave to other workgroups of
// the same grid (the dense pose unit's solves) where a coarser slicing allows it -- any slicing adds in the same tile order
bool cov_loss_tiled_shape(int B, int N, int* T, int* S, int* TS, int reserved_groups = 0);

struct PnpParams {
    const float* K;       // (B,3,3)
    const float* pts2d;   // (B,Nmax,2)
    const float* pts3d;   // (B,Nmax,3)
    const float* sqrtL;   // (B,Nmax,2,2) lower factor of the 2x2 information matrix, or
    const float* sqrt_diag;  // (B,Nmax,2) its diagonal when the factor is diagonal (sqrtL == null)
    const int* counts;    // (B,) valid points per pose, or null (== Nmax)
    const float* start;   // (B,7) start poses, or null: read them from `states` (in-place form)
    float* states;        // (B,7) out: optimum if converged, else the start pose
    float* result_tr;     // (B,) final trust-region radius
    int* rets;            // (B,) 0 ok / 1 invalid (between a split launch and its rescue launch also kPnpPartNeverArrived)
    int* iters;           // (B,) LM iterations used, or null
    int B, Nmax, max_iter;
    float ftol;
    double* trace;        // diagnostic launch only (launch_pnp_lm_trace): (B,trace_rows,8) per-iteration rows, else null
    int trace_rows;
    int options;          // kPnp* bits (launch_pnp_lm only): what the callers otherwise do with element-wise launches in front
    const unsigned char* weight_mask;  // (B,Nmax) or null: unit information where non-zero, none elsewhere (instead of sqrtL / sqrt_diag)
    int pose_mod;         // > 0: K and start have pose_mod rows, pose b reads row b % pose_mod (start must be given)
    void* split_ws;       // launch_pnp_lm only, or null: pnp_split_workspace_bytes(B, Nmax) bytes, zeroed once -> split_parts workgroups per pose
    int split_parts;      // pnp_split_parts(B, Nmax) when split_ws is given
};
constexpr int kSplitMinPoints = 2048;
// rets[b] of a split launch whose part 0 waited in vain for another part: "not solved yet", as opposed to 1 = "did not converge"
// (ceres.cpp:134-138).  launch_pnp_lm always follows a split launch by the rescue launch, which re-solves exactly these poses, so the value
// never reaches a caller.
constexpr int kPnpPartNeverArrived = 2;
int device_compute_units();          // of the current device, cached
int split_parts_for(int units);      // 8 / 4 / 2 workgroups per unit of work while units x parts fits the device's compute units, else 1  // rows up to here: one workgroup per pose (the exchange between the parts would cost more than it saves)
int pnp_split_parts(int B, int Nmax);  // workgroups per pose of the split form: 1 (not worth it / grid would not be resident at once), 2, 4, 8
size_t pnp_split_workspace_bytes(int B, int Nmax);
enum PnpOptions { kPnpWeightsAreIcov = 1,  // sqrt_diag holds inverse VARIANCES: take the square root at the load (cer_solver.py:33-36)
                  kPnpNanToNum = 2,        // torch.nan_to_num on K, points, weights and start at the load (cer_solver.py:29-31)
                  kPnpWeightsAreStd = 4 }; // (with kPnpWeightsAreIcov) sqrt_diag holds standard DEVIATIONS: 1 / (s s) first (test.py:52 `1/(pts2d_std**2)`, torch's own
                                           // float operations), then as for inverse variances
int launch_pnp_lm(const PnpParams& p, hipStream_t stream);
// launch_pnp_lm(a) then launch_pnp_lm(b) -- as one launch where the shapes allow (include/lc_amd.h: lc_pnp_lm_chain_f32)
int launch_pnp_lm_chain(const PnpParams& a, const PnpParams& b, hipStream_t stream);
int launch_pnp_lm_trace(const PnpParams& p, hipStream_t stream);  // same solve + p.trace rows (parity diagnostics, not a hot path)
// both of the above in one grid (N <= 64 only; returns 3 otherwise)
int launch_pose_unit(const LossParams& lp, const PnpParams& pp, hipStream_t stream);
// the same for dense shapes (256 < N <= 2048, lp.workspace given): tiled loss workgroups and four-wave solve workgroups in one grid;
// 3: shape not supported / workspace too small
int launch_pose_unit_dense(const LossParams& lp, const PnpParams& pp, hipStream_t stream);

enum HeadDtype { kHeadF32 = 0, kHeadF16 = 1, kHeadBF16 = 2 };  // element type of the (M,H,W) maps (input and its gradient)

struct HeadParams {
    const void* in;       // (M,H,W) logits (or probabilities when is_prob), element type `dtype`
    float* mean;          // (M,2) x,y
    float* std;           // (M,2)
    float* stats;         // (M,4): lse (or sum p), var_x, var_y, unused -- saved for backward
    int M, H, W;
    int is_prob;
    int dtype;
    int variant;          // memory-policy bits, filled by the launcher
};
int launch_head_fwd(const HeadParams& p, hipStream_t stream);

struct HeadBwdParams {
    const void* in;        // (M,H,W), element type `dtype`
    const float* mean;     // (M,2)
    const float* std;      // (M,2)
    const float* stats;    // (M,4)
    const float* g_mean;   // (M,2)
    const float* g_std;    // (M,2)
    void* g_in;            // (M,H,W), element type `dtype`
    int M, H, W;
    int is_prob;
    int dtype;
    int variant;           // memory-policy bits, filled by the launcher
};
int launch_head_bwd(const HeadBwdParams& p, hipStream_t stream);

struct RansacParams {
    const float* K;        // (B,3,3)
    const float* pts3d;    // (B,Nmax,3)
    const float* pts2d;    // (B,Nmax,2)
    const int* counts;     // (B,) or null
    const float* reproj_err_per_pose;  // (B,) or null (use reproj_err); see per_pose_divides
    float* states;         // (B,7) out
    unsigned char* inlier_mask;  // (B,Nmax) out
    int* n_inliers;        // (B,) out
    int* invalid;          // (B,) out
    int B, Nmax, rounds;
    float reproj_err;
    unsigned seed;
    int* best_hyp;         // (B,) out or null: index of the winning hypothesis (-1 when invalid) -- parity diagnostics
    int* valid_counts;     // (B,) out or null: the pose's point count, 0 when invalid -- the `counts` of a refinement that must skip failed poses
    void* workspace;       // null: single launch (one workgroup per pose); else the split form (hypotheses / scoring / selection)
    size_t workspace_bytes;
    // Optional second-stage selection (test.py:129-133, 'weighted-filtered'): the correspondences the winner calls inliers, compacted
    // to the front of their rows by the workgroup that writes the inlier mask -- what lc_dense_select_f32 in 'mask' mode would make
    // of that mask in a launch of its own.  sel_w == null: none.
    const float* sel_w;        // (B,Nmax,2) the weights that travel with the correspondences
    const int* sel_in_index;   // (B,Nmax) source index of each input entry, or null (identity)
    float* sel_pts2d;          // (B,Nmax,2) out
    float* sel_w_out;          // (B,Nmax,2) out
    float* sel_pts3d;          // (B,Nmax,3) out
    int* sel_index;            // (B,Nmax) out or null
    int* sel_counts;           // (B,) out
    int sel_min_count;
    unsigned sel_seed;
    int ticketed;              // split form only: scoring and selection in ONE launch, the last workgroup of a pose to finish selects
    int pose0;                 // index of the batch's first pose in the caller's numbering: the hypothesis stream and the padding draw of pose b are
                               // those of pose pose0 + b, so a batch solved in several sub-batches (concurrently, on several streams) gives the
                               // results of the one call
    int per_pose_divides;      // 0: reproj_err_per_pose[b] IS the threshold of pose b (lc_pnp_ransac_init .. init4); 1 (init5 with reproj_err > 0):
                               // the threshold is reproj_err / reproj_err_per_pose[b], a non-positive divisor leaving reproj_err itself
};
int launch_pnp_ransac(const RansacParams& p, hipStream_t stream);  // 3: workspace too small
size_t pnp_ransac_workspace_bytes(int B, int Nmax, int rounds);
void pnp_ransac_workspace_layout(int B, int Nmax, int rounds, size_t out[6]);  // byte offsets of hyp64, hyp32, partials; H, C; total bytes

struct BitsParams {
    const void* logits;           // (B,C,H,W) code logits of element type map_dtype (lc_map.h), C = bits[0]+bits[1]+bits[2]
    const unsigned char* gt_bits; // (B,C,H,W) raw ground-truth bits (training decode) or null
    const unsigned char* gt_msk;  // (B,H,W) object mask or null (all inside)
    const float* g_out;           // (B,N,3) cotangent (backward) or null
    float* out;                   // fwd: (B,N,3) normalised coordinates
    void* d_logits;               // bwd: (B,C,H,W), element type map_dtype
    int B, C, H, W, N, top, left, sample;
    int bits[3];
    int black_factor;             // -1: black background (default), +1 otherwise
    // training decode only: normalised coordinates -> object coordinates in the kernel (losses.py:17-47 nn_out_to_xyz, :163-184):
    const float* out_scale;       // (B,3) noc_scale, or null: out = noc * scale
    const float* out_xform;       // (B,4,4) model transform T, or null: out = (noc * scale - T[:3,3]) @ T[:3,:3]
    int out_planar;               // inference decode only: write (B,3,H,W) planes (what the dense front end reads) instead of (B,H,W,3)
    int map_dtype;                // kMapF32 / kMapF16 / kMapBF16: element type of logits and d_logits (outputs and cotangents are fp32)
    long long logits_bs;          // elements between consecutive samples of `logits` (C*H*W for a dense batch; larger for a channel slice of the
                                  // network's (B,C_all,H,W) output, ptnet.py:56: no copy in front of the kernel); d_logits is dense
    // inference decode of SELECTED pixels only (launch_bits_decode_rows): entry k < rows_counts[b] of row b is sampled pixel rows_index[b][k] of the
    // (top, left, sample) grid; its object coordinates go to out[(b * rows_N + k) * 3 ..]
    const int* rows_index;        // (B, rows_N)
    const int* rows_counts;       // (B,)
    int rows_N;
    int tile_groups;              // set by launch_bits_decode_gt_bwd for its tile-shaped kernel: thread groups that share a tile's channels
};
int launch_bits_decode_gt_fwd(const BitsParams& p, hipStream_t stream);
int launch_bits_decode_gt_bwd(const BitsParams& p, hipStream_t stream);
int launch_bits_decode(const BitsParams& p, hipStream_t stream);
int launch_bits_decode_rows(const BitsParams& p, hipStream_t stream);

struct MetricsParams {
    const float* R_est;  // (B,3,3)
    const float* t_est;  // (B,3)
    const float* R_gt;   // (B,3,3)
    const float* t_gt;   // (B,3)
    const float* pts;    // (P,3) model vertices, all objects back to back
    const int* pts_off;  // (B,) first vertex of the pose's object, or null (0)
    const int* pts_cnt;  // (B,) vertex count of the pose's object, or null (M)
    float* out;          // (B,4) adi, add, re [deg], te
    int B, M, want_adi;
};
int launch_pose_errors(const MetricsParams& p, hipStream_t stream);

struct DenseParams {
    const void* xyz;         // (B,3,H,W) network xyz head        } element type map_dtype (lc_map.h)
    const void* wlogits;     // (B,2,H,W) weight logits           }
    const void* wscale;      // (B,) per-sample weight scale, element type wscale_dtype
    const float* noc_scale;  // (B,3) or null
    float* pts2d;            // (B,N,2)
    float* inv_std;          // (B,N,2)
    float* pts3d;            // (B,N,3)
    float* lse;              // (B,) saved for backward
    int B, H, W, N, top, left, sample;
    const void* vis_logits;  // (B,H,W) visibility logits (element type map_dtype) or null: test-time selection mask of the sampled pixels (test.py:88-90)
    float vis_thresh;        //   sigmoid(logit) > vis_thresh
    unsigned char* vis_mask; // (B,N) out (with vis_logits)
    int map_dtype;           // kMapF32 / kMapF16 / kMapBF16; the (B,N,.) rows are fp32 whatever the maps are
    int wscale_dtype;        // element type of wscale (fp32 under autocast -- exp is an fp32 op -- or the model's 16-bit type)
    int xyz_dtype;           // element type of xyz alone: map_dtype, or kMapF32 next to 16-bit logits (test time, binary-code heads: the decoded
                             // coordinate planes are fp32 whatever the network's type is)
    long long xyz_bs, wl_bs, vis_bs;  // elements between consecutive samples of xyz / wlogits / vis_logits (3HW, 2HW, HW for dense batches;
                             // larger for channel slices of the network's (B,C_all,H,W) output, ptnet.py:56: consumed in place)
};
// batch strides left at 0 mean a dense batch
inline DenseParams with_dense_strides(DenseParams p) {
    const long long HW = (long long)p.H * p.W;
    if (!p.xyz_bs) p.xyz_bs = 3 * HW;
    if (!p.wl_bs) p.wl_bs = 2 * HW;
    if (!p.vis_bs) p.vis_bs = HW;
    return p;
}
int launch_dense_fwd(const DenseParams& p, hipStream_t stream);

struct DenseBwdParams {
    const void* wlogits;     // (B,2,H,W), element type map_dtype
    const void* wscale;      // (B,), element type wscale_dtype
    const float* noc_scale;  // (B,3) or null
    const float* lse;        // (B,)
    const float* g_inv_std;  // (B,N,2) or null
    const float* g_pts3d;    // (B,N,3) or null
    void* d_xyz;             // (B,3,H,W) or null  } element type map_dtype: the gradient of a map in the map's own type
    void* d_wlogits;         // (B,2,H,W) or null  }
    void* d_wscale;          // (B,) or null, element type wscale_dtype
    int B, H, W, N, top, left, sample;
    int map_dtype;
    int wscale_dtype;
    long long wl_bs;         // elements between consecutive samples of wlogits (the gradient maps are dense)
};
int launch_dense_bwd(const DenseBwdParams& p, hipStream_t stream);

constexpr int kDenseAuxMaxBlocks = 1024;  // rows of the caller-provided `partials` workspace of lc_dense_aux_fwd_f32 (3 doubles each)
struct DenseAuxParams {
    const void* xyz;                  // (B,3,HW) xyz head (element type map_dtype), or null (binary-code heads have no loss_noc)
    const unsigned char* msk_noc_u8;  // (B,HW) object mask as bool bytes, or
    const float* msk_noc_f32;         // (B,HW) the same as floats (exactly one of the two with xyz)
    const float* noc_tgt;             // (B,3,HW)
    const void* seg_logits;           // (B,HW) msk_vis_logits, element type map_dtype
    const float* msk_vis;             // (B,HW) visibility target
    const void* wlogits;              // (B,2,HW) xyz_weight_logits (element type map_dtype), or null (no warm-up blend)
    int seg_type;                     // 0: binary_cross_entropy_with_logits, 1: Loss_seg_L1
    float* losses;                    // (3) out: loss_noc, loss_seg, loss_weight_seg (forward)
    double* partials;                 // (kDenseAuxMaxBlocks,3) workspace (forward)
    unsigned* ticket;                 // kArrivalWords (lc_common.h), zero between launches (forward)
    const float* g_noc;               // upstream cotangents: device scalars or null (backward)
    const float* g_seg;
    const float* g_wseg;
    void* d_xyz;                      // (B,3,HW) or null (backward)   } element type map_dtype
    void* d_seg;                      // (B,HW) or null                }
    void* d_wlogits;                  // (B,2,HW) or null              }
    int B, HW;
    int map_dtype;                    // of the network outputs (xyz, seg_logits, wlogits) and their gradients; targets and masks stay fp32 / bytes
    long long xyz_bs, seg_bs, wl_bs;  // elements between consecutive samples of xyz / seg_logits / wlogits (channel slices in place); gradients dense
};
int launch_dense_aux_fwd(const DenseAuxParams& p, hipStream_t stream);
int launch_dense_aux_bwd(const DenseAuxParams& p, hipStream_t stream);

constexpr int kBinMaxChannels = 128;  // code bits of Loss_xyz_bin (3 axes x up to 24 bits)
struct BinLossParams {
    const void* logits;             // (B,C,HW) code logits, element type map_dtype
    const unsigned char* gt_bits;   // (B,C,HW) target bits as bool bytes
    const void* msk_vis_logits;     // (B,HW), element type map_dtype
    float* histogram;               // (C) EMA of the per-bit Hamming error rate: read AND updated (forward)
    float momentum;
    float* loss;                    // (1) out (forward)
    float* bin_weights;             // (C) out (forward), in (backward)
    double* partials;               // (C * 32, 3) workspace (forward)
    unsigned* ticket;               // kArrivalWords (lc_common.h), zero between launches (forward)
    // sharded form (the batch is split over ranks, the histogram is the whole batch's): forward with counts_out != null stops after the
    // counts; the caller all-reduces them; launch_xyz_bin_loss_finish takes counts_in and does the rest
    long long* counts_out;          // (C + 1) out: per-bit Hamming errors inside the hard visibility mask, then the mask's population
    const long long* counts_in;     // (C + 1) in (finish)
    float* bce_mean;                // (C) this rank's per-bit BCE means: out of the counts launch, in to the finish launch
    const float* g_loss;            // device scalar (backward)
    void* d_logits;                 // (B,C,HW) (backward), element type map_dtype
    int B, C, HW;
    int vec;                        // HW % 4 == 0 and all maps 16-byte (bits: 4-byte) aligned: four pixels per request
    int chunks;                     // workgroups per code channel (set by the launcher)
    int map_dtype;
    long long logits_bs, vis_bs;    // elements between consecutive samples of logits / msk_vis_logits (channel slices in place); d_logits dense
};
int launch_xyz_bin_loss_fwd(const BinLossParams& p, hipStream_t stream);  // 3: more than kBinMaxChannels bits
int launch_xyz_bin_loss_finish(const BinLossParams& p, hipStream_t stream);  // reads counts_in, bce_mean, histogram; writes histogram, bin_weights, loss
int launch_xyz_bin_loss_bwd(const BinLossParams& p, hipStream_t stream);

constexpr int kClipMaxBlocks = 512;  // length of the caller-provided `partials` workspace of lc_sqnorm_f32
struct ClipParams {
    const void* x;       // gradient, element type dtype (lc_map.h: a 16-bit head's gradient is clipped where it lies, fp32 arithmetic)
    long long n;
    int vec;             // x (and out) 16-byte aligned: float4 stream
    // lc_sqnorm
    double* partials;    // (kClipMaxBlocks,) workspace
    unsigned* ticket;    // zero-initialised counter, left at zero
    float* sq;           // device scalar: sum of squares (accumulated into when `accumulate`)
    int accumulate;
    float* state_snapshot;   // optional: receives *state_in (the running maximum before this hook call)
    // lc_clip_apply
    const float* state_in;   // max_norm before the call (<= 0: not started)
    float initial_max_norm, scale, keep, gain;  // keep = 1 - momentum, gain = momentum * scale
    void* out;           // element type dtype
    float* state_out;    // max_norm after the call, or null (do not update)
    float* norm_out;     // total norm, or null
    int dtype;           // kMapF32 / kMapF16 / kMapBF16
};
int launch_sqnorm(const ClipParams& p, hipStream_t stream);
int launch_clip_apply(const ClipParams& p, hipStream_t stream);

struct KptParams {
    const float* K;      // (B,3,3)
    const float* pose;   // (B,7) wxyz + xyz
    const float* pts3d;  // (B,N,3)
    const float* pts2d;  // (B,N,2)
    const float* std;    // (B,N,2) predicted sigma
    float* nll;          // (B,) sum over the sample's N*2 terms
    float* d_pts2d;      // (B,N,2) d nll[b] / d pts2d, or null
    float* d_std;        // (B,N,2) d nll[b] / d sigma, or null
    int B, N;
};
int launch_kpt_nll(const KptParams& p, hipStream_t stream);

struct SelectParams {
    const float* pts2d;         // (B,N,2)
    const float* inv_std;       // (B,N,2) weights
    const float* pts3d;         // (B,N,3)
    const unsigned char* mask;  // (B,N) segmentation / inlier mask (modes 0 and 2), else null
    const int* in_counts;       // (B,) valid prefix of each input row, or null (N)
    const int* in_index;        // (B,N) source index of each input entry (second-stage selection), or null (identity)
    float* o_pts2d;             // (B,N,2) compacted
    float* o_w;                 // (B,N,2) compacted weights (squared when `square`)
    float* o_pts3d;             // (B,N,3) compacted
    int* o_index;               // (B,N) source indices of the survivors, or null
    int* counts;                // (B,)
    int B, N;
    int mode;                   // 0 mask, 1 quantile, 2 quantile_in_mask (test.py:94-104)
    float quantile, one_minus_q;
    int square, min_count;
    unsigned seed;
    int pose0;                  // as RansacParams::pose0: the padding draw of row b is that of row pose0 + b
    void* split_ws;             // launch_dense_frontend_select only, or null: dense_select_split_workspace_bytes(B, N) bytes, zeroed once.  Seen by the
                                // one-workgroup kernel it means "rescue role": only objects whose region is marked are selected (lc_select.hip)
    int split_parts;            // set by the launch: workgroups per object (dense_select_split_parts), 1 = one workgroup per object
};
int dense_select_split_parts(int B, int N);  // 1 (rows of up to 4096 candidates / more than 128 objects: the grid would not be resident at once), 2, 4, 8
size_t dense_select_split_workspace_bytes(int B, int N);
int launch_dense_select(const SelectParams& p, hipStream_t stream);
// front end + selection in one launch (test time, N <= 1024): the input arrays of `p` are unused (null), `d` names the maps
int launch_dense_frontend_select(const SelectParams& p, const DenseParams& d, hipStream_t stream);  // 3: N > 1024

}  // namespace lc
