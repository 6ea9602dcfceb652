/*
 * lc_amd -- C ABI of the MI355X (gfx950) hot path of fulliu/lc.
 *
 * One shared library (liblc_amd.so, built from lc_amd/csrc/ by `python __graft_entry__.py build`) exports
 *   (1) the reference's OWN native entry point, symbol-for-symbol, so the reference's cffi binding
 *       (lib/pnp/pnp_ceres.py:93-140) can load this library instead of its Ceres extension, and
 *   (2) device-pointer entry points -- ONE per operation -- for the fused kernels, which the Python host layer (lc_amd/ *.py,
 *       mirroring lib/cov_mixed.py, lib/pnp/cer_solver.py, ptnet.py, losses.py, floatbits.py) binds with ctypes.
 * Names keep the generation suffix they had when they superseded an earlier form (lc_pnp_lm3_f32, lc_cov_loss3_fwd_bwd_f32, ...): the
 * earlier forms are not exported any more, every one of them is the surviving call with NULL / 0 in the arguments it lacked.
 * Plain pointers and sizes only; no torch types.  All `float*`/`int*` of the device API are DEVICE pointers,
 * `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on that stream.
 * Return value: 0 on success, non-zero on error (lc_amd_last_error() gives the text).
 * Alignment: rows of two floats (pts2d, inv_std / sqrt_diag / std and their gradients) must be 8-byte aligned, 2x2 factors
 * (sqrtL) and maps whose width is a multiple of four 16-byte aligned -- true for every contiguous batch and every row
 * slice of one; a misaligned pointer is rejected (error 1), never read through a narrower fallback silently.
 */
#ifndef LC_AMD_H
#define LC_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_AMD_VERSION 2 /* 2: one entry point per operation (the superseded generations of version 1 are gone) */
#define LC_AMD_LOSS_AUX_STRIDE 40 /* per sample: prior_error, cov_err, linear_err, not_spd, Hinv[36] */

int lc_amd_version(void);
const char *lc_amd_last_error(void);
/* sha256 (hex) of the .hip/.h sources this library was compiled from ("unrecorded" for a build that did not pass it): the
 * loader rebuilds a library whose sources have changed (lc_amd/build.py reads the same string from the file's bytes) */
const char *lc_amd_source_hash(void);

/* ------------------------------------------------------------------------------------------------
 * (1) Reference ABI -- replaces /root/reference/lib/pnp/cxx/ext.h:2-15 (implemented in
 *     lib/pnp/cxx/ceres.cpp:147-177 on top of Ceres 2.1.0).  HOST pointers, arrays-of-pointers, one entry per job:
 *       init_states[i] : 7 floats w,x,y,z,tx,ty,tz -- updated IN PLACE only when rets[i]==0 (ceres.cpp:134-144)
 *       cam_Ks[i]      : row-major 3x3, first 6 floats read (ceres.cpp:99-101)
 *       pts2ds[i]      : ptCnts[i] x 2,  pts3ds[i] : ptCnts[i] x 3
 *       icov_sqrtLs[i] : ptCnts[i] x 2 x 2 row-major lower factor, element [0][1] ignored (ceres.cpp:25-27)
 *       rets[i]        : 0 ok / 1 invalid (fewer than 3 points, or the solve did not end in CONVERGENCE)
 *       result_trs[i]  : final trust-region radius (1 when skipped)
 *     Caller owns every buffer; nothing outlives the call; no exceptions/errno.  num_threads is accepted for
 *     signature compatibility and ignored: all jobs run concurrently on the GPU, one wavefront per job.
 *     If no GPU can be used the call prints to stderr and marks every job invalid (there is NO CPU fallback).
 * ------------------------------------------------------------------------------------------------ */
void pnp_ceres_f32_omp(float **init_states, float **cam_Ks, float **pts2ds, float **pts3ds, float **icov_sqrtLs,
                       int *ptCnts, int maxIterCnt, float function_tolerance, int printSummary, float *result_trs,
                       int *rets, int job_count, int num_threads);

/* ------------------------------------------------------------------------------------------------
 * (2a) Batched weighted PnP on device-resident, zero-padded batches -- the GPU form of what
 *      lib/pnp/cer_solver.py:22-44 builds before calling the solver (so no device->host round trip).
 *      K (B,3,3)  pts3d (B,Nmax,3)  pts2d (B,Nmax,2)  counts (B) or NULL (= Nmax)
 *      exactly one of: sqrtL (B,Nmax,2,2) lower factor | weights_diag (B,Nmax,2) its diagonal (cer_solver.py:37-40) | weight_mask
 *      start (B,7) or NULL; states (B,7): with start==NULL (or start==states) the reference's in-place rule applies
 *      (states holds the start pose on entry and is overwritten only for converged jobs); with a separate start,
 *      states is output only: the optimum for converged jobs, a copy of start otherwise (cer_solver.py:51-52).
 *      result_tr (B), rets (B), iters (B) or NULL
 *  The element-wise work of the callers is folded into the load (each was a separate launch in front of the solve):
 *   weights_diag + LC_PNP_WEIGHTS_ARE_ICOV: the (B,Nmax,2) tensor holds inverse VARIANCES; their square root is the information
 *       factor (lib/pnp/cer_solver.py:33-36 `icovs.sqrt()`);
 *   LC_PNP_NAN_TO_NUM: torch.nan_to_num (NaN -> 0, +-inf -> +-FLT_MAX) on K, pts3d, pts2d, the weights and start
 *       (cer_solver.py:29-31 `filter_input_nan`); an invalid job returns the FILTERED start;
 *   weight_mask (B,Nmax) uint8: unit information on the flagged correspondences, none on the others (the RANSAC inlier
 *       refinement of lc_amd/pnp/gpu_solver.py);
 *   pose_mod > 0: K and start have pose_mod rows and pose b reads row b % pose_mod -- several solves of the same objects on
 *       different correspondence selections (test.py:120,133 'weighted' and 'weighted-filtered') as ONE launch of B = k pose_mod poses.
 *  options = 0, pose_mod = 0, workspace = NULL: the plain solve.
 * ------------------------------------------------------------------------------------------------ */
#define LC_PNP_WEIGHTS_ARE_ICOV 1
#define LC_PNP_NAN_TO_NUM 2
/* with LC_PNP_WEIGHTS_ARE_ICOV: weights_diag holds the predicted standard DEVIATIONS of the sparse head (test.py:52 `inv_cov2d = 1/(pts2d_std**2)`):
 * 1 / (s * s) is formed at the load with the float operations torch uses, then filtered and rooted as inverse variances are */
#define LC_PNP_WEIGHTS_ARE_STD 4

/* Batches of FEW poses with THOUSANDS of correspondences each (the test-time solves behind the dense heads,
 * test.py:120-133 with 64 objects x ~3000 selected pixels): given a workspace, such a batch is solved by several workgroups per pose
 * -- each sums its share of the correspondences, the partial normal equations meet in the workspace, and all of them take the same
 * LM steps -- instead of one workgroup per pose on a quarter of the chip.
 *   lc_pnp_lm_workspace_bytes(B, Nmax): bytes that shape needs; 0 when it is solved by one workgroup per pose anyway (Nmax <= 2048,
 *       or more poses than half the device's compute units -- B > 128 on an MI355X: the grid would not fit the chip at one workgroup per CU).
 *   workspace: that many bytes, 128-byte aligned, ZEROED ONCE by the caller; after that it belongs to these calls (each leaves it ready for
 *       the next of any shape on the same stream; calls that may run concurrently need a workspace each).  NULL: one workgroup per pose.
 * Results: those of the one-workgroup solve up to the order of the fp64 sums over the correspondences (tests/test_gpu_pnp_split.py); rets keeps
 * the reference's meaning -- 0 solved, 1 not usable / did not converge (ceres.cpp:134-138) -- and nothing else.
 * Scheduling: the workgroups of a pose wait for each other, and the launch is sized to one workgroup per compute unit so that they normally
 * all run at once.  Nothing DEPENDS on that: a workgroup that has waited a few milliseconds in vain stops, and the call always enqueues a
 * second launch behind the first that (a) re-zeroes the exchange region of every pose a workgroup gave up on and (b) solves those poses
 * with one workgroup each, adding the partial sums in the order the several workgroups would have -- the SAME bits.  Other streams, other
 * processes or a CU mask holding compute units therefore cost time, never a pose, and never change a result
 * (tests/test_gpu_contention.py).  What a C caller must do: zero the workspace once; give concurrent calls (different streams) a workspace
 * each; nothing else.  Callers that overlap many solves on several streams may still prefer workspace = NULL (one workgroup per pose):
 * contended split launches are correct but slow. */
size_t lc_pnp_lm_workspace_bytes(int B, int Nmax);
/* How often a rescue launch had to recompute a unit over the life of a split-form workspace: kind 0 = lc_pnp_lm3_f32's, 1 =
 * lc_dense_frontend_select3's.  A non-zero count says the launches were contended (results are unaffected, time is not): such a caller
 * does better with workspace = NULL.  Synchronises `stream`, reads the counters off the device (a diagnostic, not for a hot loop);
 * -1 on error. */
long long lc_split_workspace_rescues(const void *workspace, size_t workspace_bytes, int kind, void *stream);
int lc_pnp_lm3_f32(const float *K, const float *pts3d, const float *pts2d, const float *sqrtL, const float *weights_diag,
                   const unsigned char *weight_mask, const int *counts, const float *start, float *states, float *result_tr,
                   int *rets, int *iters, int B, int Nmax, int max_iter, float function_tolerance, int options, int pose_mod,
                   void *workspace, size_t workspace_bytes, void *stream);

/* Two solves, the second starting where the first ends -- the RANSAC inlier refinement followed by the weighted solve(s) of
 * test.py:120,133 -- as ONE call.  A job is the argument list of lc_pnp_lm3_f32 up to pose_mod; the call is defined as
 *     lc_pnp_lm3_f32(first ..., workspace, workspace_bytes, stream);  lc_pnp_lm3_f32(second ..., workspace, workspace_bytes, stream);
 * and returns what those return.  Where the shapes allow (both 256 < Nmax <= 1024, second->B a multiple of first->B, and
 * second->start either unrelated to first->states or reading its row b % first->B: pose_mod == first->B, or pose_mod == 0 with equal
 * B) it is ONE launch: workgroup b solves pose b % first->B of the first job, then pose b of the second -- a pose of the first job
 * that several second-stage poses start from is solved by each of them (same inputs, same arithmetic, same bits; its outputs are
 * written by each with identical values).  Same results as the two calls bit for bit (tests/test_gpu_pnp.py).  Apart from
 * second->start == first->states the buffers of the two jobs must not overlap. */
typedef struct lc_pnp_lm_job {
    const float *K, *pts3d, *pts2d, *sqrtL, *weights_diag;
    const unsigned char *weight_mask;
    const int *counts;
    const float *start;
    float *states, *result_tr;
    int *rets, *iters;
    int B, Nmax, max_iter;
    float function_tolerance;
    int options, pose_mod;
} lc_pnp_lm_job;
/* workspace (NULL: none) for jobs that take the split form: max over the two jobs of lc_pnp_lm_workspace_bytes(B, Nmax) bytes */
int lc_pnp_lm_chain2_f32(const lc_pnp_lm_job *first, const lc_pnp_lm_job *second, void *workspace, size_t workspace_bytes, void *stream);

/* (2a') Parity diagnostics of (2a): the same solve (same template body, so the same arithmetic) that also records the
 *      trust-region schedule -- what `Solver::Summary::iterations` holds after ceres::Solve (ceres.cpp:126-130) --
 *      into trace (B,trace_rows,8) doubles, one row per iteration i < trace_rows:
 *        [kind (0 invalid step | 1 accepted | 2 rejected | 3 parameter tolerance | 4 function tolerance),
 *         cost at x, candidate cost, model cost change, relative decrease, ||step||, radius after, max|gradient| after].
 *      The caller zero-fills trace.  Not a hot path (separate, slower kernel). */
int lc_pnp_lm_trace_f32(const float *K, const float *pts3d, const float *pts2d, const float *sqrtL,
                        const float *sqrt_diag, const int *counts, const float *start, float *states, float *result_tr,
                        int *rets, int *iters, int B, int Nmax, int max_iter, float function_tolerance, double *trace,
                        int trace_rows, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2b) Linear-covariance loss, forward + backward in one launch -- replaces the autograd graph of
 *      lib/cov_mixed.py:100-150 Loss_cov_mixed (cov_2d=False).  K (B,3,3) pose (B,7) pts3d (B,N,3) pts2d (B,N,2)
 *      inv_std (B,N,2) valid (B,N)|NULL bbox_3d (B,8,3) grad_out (B)|NULL(=1)
 *      -> loss (B); d_pts2d,d_inv_std (B,N,2) (both NULL = forward only); d_pts3d (B,N,3)|NULL; aux (B,40)|NULL
 *      Gradients are d(sum_b grad_out[b]*loss[b]) / d(input).  cov_2d: the switch of cov_mixed.py:111,125-130 (covariance of the
 *      PROJECTED bbox corners; no reference call site enables it, losses.py:333,383).
 *      Declared below as lc_cov_loss3_fwd_bwd_f32 (with the optional workspace of the dense shapes).
 * ------------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------------
 * (2b') One "pose unit" per pose in ONE launch: (2b) on B samples and (2a) on the same B correspondence sets
 *      (diagonal information factor pnp_sqrt_diag (B,N,2), start poses pnp_start (B,7) -> pnp_states (B,7)), N <= 64.
 *      The two computations are independent; sharing a grid fills the chip at small B (bench.py's headline step).
 * ------------------------------------------------------------------------------------------------ */
/* N <= 64: one wavefront per pose for each half (workspace unused: NULL, 0).  The dense shapes as well: 256 < N <= 2048 with
 * workspace = lc_cov_loss_workspace_bytes(B, N) > 0 bytes (256-byte aligned like lc_cov_loss3_fwd_bwd_f32's, zero-filled once, left zero): the tiled loss's
 * workgroups and the four-wave solve's share one grid (B = 32, N = 1024: 128 + 32 workgroups at the same time instead of two launches
 * back to back).  Other shapes: return code 3 -- launch lc_cov_loss3_fwd_bwd_f32 and lc_pnp_lm3_f32 separately.  pnp_iters (B)|NULL.
 * Results bit for bit those of the two stand-alone launches. */
int lc_pose_unit2_f32(const float *K, const float *pose, const float *pts3d, const float *pts2d, const float *inv_std,
                      const float *valid, const float *bbox_3d, const float *grad_out, int B, int N, float max_err_len,
                      float rel_thresh, float w_e_thresh, float *loss, float *d_pts2d, float *d_inv_std, float *d_pts3d,
                      const float *pnp_sqrt_diag, const float *pnp_start, float *pnp_states, float *pnp_result_tr,
                      int *pnp_rets, int *pnp_iters, int pnp_max_iter, float pnp_function_tolerance, void *workspace,
                      size_t workspace_bytes, void *stream);

/* (2b'') The loss kernel (2b).  A caller workspace lets the dense shapes (N > 256: configs/glmo.yaml N = 1024, zlmo N = 1849;
 * losses.py:336-386) spread ONE sample over several compute units: the sample's 64-correspondence tiles are dealt to 256-thread
 * workgroups (4, 8 or 16 tiles each) that exchange the partial sums of the normal equations once through the workspace.
 * Results are bit-identical to the workspace-less call (all forms add the per-sample sums in the same tile order).
 *   lc_cov_loss_workspace_bytes(B, N): bytes needed, 0 when the shape would not use it (N <= 256, or a batch that fills the
 *   chip anyway).  The caller zero-fills the workspace ONCE per (B, N); every launch leaves it zeroed except word 2 of its header,
 *   which counts hand-offs that timed out (none can, by construction); a sample any of whose workgroups timed out returns loss = NaN
 *   (its gradients are then undefined): re-zero the workspace after that.  One workspace serves the launches of ONE stream (or of graphs replayed
 *   one at a time); concurrent launches need one each.  workspace == NULL: one workgroup per sample. */
size_t lc_cov_loss_workspace_bytes(int B, int N);
int lc_cov_loss3_fwd_bwd_f32(const float *K, const float *pose, const float *pts3d, const float *pts2d,
                             const float *inv_std, const float *valid, const float *bbox_3d, const float *grad_out,
                             int B, int N, float max_err_len, float rel_thresh, float w_e_thresh, int cov_2d, float *loss,
                             float *d_pts2d, float *d_inv_std, float *d_pts3d, float *aux, void *workspace,
                             size_t workspace_bytes, void *stream);

/* dst[b, :] = scale[b] * src[b, :]  for up to three (B,row_len) tensors in one launch (autograd's chain-rule step) */
int lc_scale_rows_f32(const float *scale, int B, const float *src0, float *dst0, int len0, const float *src1, float *dst1,
                      int len1, const float *src2, float *dst2, int len2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2c) Sparse keypoint head -- replaces ptnet.py:59-66 + ptnet.py:85-115.  in (M,H,W): logits (is_prob=0; the
 *      spatial softmax of ptnet.py:61 is fused) or probabilities (is_prob=1; ptnet.softargmax_2d_std itself).
 *      -> mean (M,2) [x,y], std (M,2), stats (M,4) saved for backward.
 * ------------------------------------------------------------------------------------------------ */
/* `in` and `g_in` are (M,H,W) of `dtype` -- fp32, or the element type a mixed-precision backbone emits (BASELINE.json configs 3, 5); statistics, mean/std and their cotangents stay fp32, arithmetic is fp32, the gradient is rounded
 * to nearest even into the map's type.  16-bit maps halve the HBM bytes of this bandwidth-bound pair. */
#define LC_F32 0
#define LC_F16 1
#define LC_BF16 2
int lc_softargmax2d_fwd(const void *in, int dtype, int M, int H, int W, int is_prob, float *mean, float *std, float *stats,
                        void *stream);
int lc_softargmax2d_bwd(const void *in, int dtype, const float *mean, const float *std, const float *stats,
                        const float *g_mean, const float *g_std, int M, int H, int W, int is_prob, void *g_in,
                        void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2d) Dense-correspondence front end -- replaces the torch glue of losses.py:355-356 (joint softmax
 *      over all 2*H*W weight logits x per-sample scale) + losses.py:142-161 dense_pnp_matching_from_xyz (strided
 *      sub-sampling with phase (top,left), noc_scale multiply, (N,C) transposes).
 *      xyz (B,3,H,W) wlogits (B,2,H,W) wscale (B) noc_scale (B,3)|NULL -> pts2d,inv_std (B,N,2) pts3d (B,N,3) lse (B)
 *      with N = ceil((H-top)/sample) * ceil((W-left)/sample).  Backward: cotangents of inv_std / pts3d (either NULL)
 *      -> d_xyz (B,3,H,W), d_wlogits (B,2,H,W), d_wscale (B) (any NULL to skip).
 *      Forward + the test-time visibility mask of the sampled pixels (test.py:88-90: sigmoid(msk_vis_logits) > seg_thresh, then the
 *      stride slice): vis_logits (B,H,W), vis_thresh -> vis_mask (B,N) uint8; both NULL = none.
 *      Declared in (2i) as lc_dense_frontend_fwd3 / lc_dense_frontend_bwd2 (any map element type, batch strides).
 * ------------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------------
 * (2g) PnP initialiser -- takes the place of lib/pnp/cv2_solver.py:69-88 (cv2.solvePnPRansac, EPnP,
 *      iterationsCount=150) in front of the weighted solve: RANSAC over P3P minimal samples, one wavefront per pose,
 *      `iterations` hypotheses rounded up to a multiple of 64.  Same zero-padded batch layout as (2a).
 *      reproj_err in pixels: the scalar.  With reproj_err_per_pose != NULL (B floats): reproj_err > 0 makes the per-pose value a DIVISOR,
 *      threshold = reproj_err / reproj_err_per_pose[b] -- test.py:56-57,115-116's `2 / gt_dict['out_pix_scale']` (rel_reproj_err) formed
 *      inside the launch, a divisor that is not positive leaves reproj_err itself; reproj_err <= 0: the per-pose value IS the threshold.
 *      -> states (B,7) w,x,y,z,tx,ty,tz; inlier_mask (B,Nmax) uint8; n_inliers (B); invalid (B) (1: fewer than 4 points
 *      or no hypothesis with >= 4 inliers; states is then the identity pose like the reference's zero rvec/tvec).
 *      best_hyp (B)|NULL: the index of the winning hypothesis of every pose (-1 when invalid).  Hypothesis h of pose b draws its four
 *      point indices from a counter-based hash of (seed, b, h) -- oracle/p3p_ransac_oracle.py restates the stream bit for bit, which
 *      makes best_hyp, n_inliers and inlier_mask integer outputs that are compared EXACTLY.
 *      Ranking of hypotheses: (inlier count, error, hypothesis index); the error that breaks count ties is the camera-plane residual
 *      at the hypothesis' depth -- sum over the inliers of |c_xy - u c_z|^2, divided by t_z^2 (c = R X + t, u the normalised pixel;
 *      division-free IEEE float32 in a fixed order: reproducible off the chip) -- not the pixel reprojection error; a hypothesis with
 *      t_z <= 0 ranks last among equal counts (error = +inf).  cv2.solvePnPRansac has no such tie-break.
 * ------------------------------------------------------------------------------------------------ */
/* Split form of the same RANSAC for batches that do not fill the chip with one workgroup per pose (64 objects x 150 hypotheses
 * x 1000+ dense correspondences): three launches -- hypotheses (one lane each), scoring (point chunks x hypotheses, spread over
 * all compute units), selection -- over a caller-provided device workspace of lc_pnp_ransac_workspace_bytes(B, Nmax, iterations)
 * bytes (16-byte aligned; contents undefined before and after; B x ceil(Nmax / 64) x hypotheses x 8 bytes of chunk partials: EVERY
 * point of a pose is sampled from and scored, as cv2.solvePnPRansac does, cv2_solver.py:72-75).  Same hypothesis stream, same per-point arithmetic and the same
 * (count, error, hypothesis index) ordering as the single launch (workspace == NULL); results do not depend on scheduling (no atomics).
 * valid_counts (B)|NULL: the pose's point count, 0 when the pose is invalid -- handed as `counts` to a following lc_pnp_lm3_f32
 * refinement it makes that solve skip the failed poses.
 * lc_pnp_ransac_workspace_layout (diagnostics: the tests that re-check the kernel's hypotheses and chunk partials on the CPU): byte
 * offsets into that workspace of out[0] the hypotheses as doubles (B,H,12: R row-major | t), out[1] the same as floats (B,H,12),
 * out[2] the chunk partials (B,C,H) of 8 bytes {inlier count int32 | error-sum float32}; out[3] = H (hypotheses, a multiple of 64),
 * out[4] = C (chunks of 64 points), out[5] = total bytes.  Returns 0, or 1 for a bad size. */
size_t lc_pnp_ransac_workspace_bytes(int B, int Nmax, int iterations);
int lc_pnp_ransac_workspace_layout(int B, int Nmax, int iterations, size_t out[6]);
/* Further options: (a) the inlier re-selection inside the selection step and (b) an optional two-launch split form.
 *  - sel_w != NULL: second-stage selection by the inlier mask (test.py:129-133, the 'weighted-filtered' solve's input) written by
 *    the workgroup that writes the mask: the inliers of every pose compacted, order kept, to the front of sel_pts2d (B,Nmax,2),
 *    sel_w_out (B,Nmax,2) <- sel_w (B,Nmax,2, the weights travelling with the correspondences), sel_pts3d (B,Nmax,3),
 *    sel_index (B,Nmax)|NULL <- sel_in_index (B,Nmax)|NULL (identity), sel_counts (B); fewer than sel_min_count inliers out of more
 *    than sel_min_count points are padded as lc_dense_select_f32 pads (test.py:108-113, sel_seed).  Bit for bit what
 *    lc_dense_select_f32(mode 0, mask = inlier_mask, in_counts = counts, in_index = sel_in_index, square = 0) returns in a launch
 *    of its own (tests/test_gpu_pnp_init_oracle.py); works with either launch form.
 *  - ticketed != 0 and workspace != NULL: TWO launches (hypotheses; scoring + selection).  One workgroup per (pose, chunk of 64
 *    points) scores, counts itself in, and the workgroup that completes the pose's count selects; nobody waits.  The chunk partials
 *    are still summed in chunk order: every output equals the three launches'.  Same workspace contract (the arrival
 *    counters in it are zeroed by the hypotheses launch).  Measured slower than the three launches on MI355X (profiles/r03/NOTES.md 8): an
 *    option for the record, not the default of the Python host side. */
/* pose_index_offset: the batch is a slice [pose_index_offset, pose_index_offset + B) of a larger one: the hypothesis stream and the padding draw of
 * pose b are those of pose pose_index_offset + b, so sub-batches solved concurrently on several streams (lc_amd/inference.py) return what the
 * one call over the whole batch returns.  lc_dense_frontend_select3 takes the same offset for its padding draw. */
int lc_pnp_ransac_init5_f32(const float *K, const float *pts3d, const float *pts2d, const int *counts, int B, int Nmax,
                            float reproj_err, const float *reproj_err_per_pose, int iterations, unsigned seed,
                            float *states, unsigned char *inlier_mask, int *n_inliers, int *invalid, int *best_hyp,
                            int *valid_counts, void *workspace, size_t workspace_bytes, int ticketed, const float *sel_w,
                            const int *sel_in_index, int sel_min_count, unsigned sel_seed, float *sel_pts2d, float *sel_w_out,
                            float *sel_pts3d, int *sel_index, int *sel_counts, int pose_index_offset, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2f) ZebraPose binary surface codes -- floatbits.py.  logits (B,C,H,W), C = n0+n1+n2 code bits
 *      (x,y,z axes back to back, floatbits.py:35-48); black_background as floatbits.py:7-11.
 *      lc_bits_decode_gt_*: floatbits.py:130-160 + :108-118 (training decode against the raw ground-truth bits gt_bits
 *      (B,C,H,W) uint8 and the object mask gt_msk (B,H,W) uint8|NULL), evaluated on the strided pixel subset
 *      (top,left,sample) of losses.py:163-184 -> noc (B,N,3); backward writes the full (B,C,H,W) logit gradient.
 *      lc_bits_decode3: floatbits.py:194-223 + :162-180 (inference Gray decode) -> noc (B,H,W,3), with the callers' coordinate map
 *      folded in (nn_out_to_xyz(..., inference=True), losses.py:17-47) and, with planar != 0, written as (B,3,H,W) planes -- what the dense
 *      front end reads: out = noc * out_scale (B,3)|NULL, then (. - T[:3,3]) @ T[:3,:3] with out_xform (B,4,4)|NULL.
 *      The training decode folds the same map in (losses.py:17-47,163-184): out (B,N,3) = (noc * out_scale - T[:3,3]) @ T[:3,:3]; the
 *      backward form takes the cotangent of `out`; out_scale = out_xform = NULL: the normalised coordinates themselves.
 *      (2d) accepts xyz = pts3d = NULL for these heads (weights / pixel grid only).  Declared in (2i).
 * ------------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------------
 * (2e) Pose-error metrics -- lib/utils/error6d.py:87-154 (add, adi, re, te) bundled as
 *      lib/utils/evaluate.py:333-339 compute_pose_errors, batched: R_* (B,3,3) t_* (B,3); pts (P,3) model vertices;
 *      pts_off/pts_cnt (B) select each pose's vertex range (both NULL: every pose uses pts[0:M]).
 *      -> out (B,4) = adi, add, re [deg], te.  want_adi=0 skips the O(M^2) nearest-neighbour search.
 * ------------------------------------------------------------------------------------------------ */
int lc_pose_errors_f32(const float *R_est, const float *t_est, const float *R_gt, const float *t_gt, const float *pts,
                       const int *pts_off, const int *pts_cnt, int B, int M, int want_adi, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2h) EMA-adaptive gradient-norm clipping -- lib/utils/grad.py:5-30 (NormClipper.clip) + :33-83 (clip_norm), the
 *      backward hook of the dense heads (losses.py:343-352,378-381), without host synchronisation:
 *        lc_sqnorm                sq (device float) = [sq +] sum x^2.  partials: LC_SQNORM_BLOCKS doubles of workspace,
 *                                 ticket: one zero-initialised unsigned (left at zero); both owned by the caller.
 *                                 state/state_snapshot (both or neither): *state_snapshot = *state, so that the apply step
 *                                 can take state_in = state_snapshot and state_out = state, i.e. update the running
 *                                 maximum IN PLACE (fixed addresses: the pair can be replayed inside a hipGraph).
 *        [all-reduce sq over the data-parallel group when the batch is sharded]
 *        lc_norm_clip_apply       norm = sqrt(sq);  limit = state_in <= 0 ? initial_max_norm : state_in;
 *                                 out = grad * min(limit / (norm + 1e-6), 1);
 *                                 state_out = state_in <= 0 ? norm*scale
 *                                           : state_in*(1-momentum) + momentum*scale*min(norm, state_in*scale)
 *                                 (state_out / norm_out may be NULL: scale a further tensor of the same hook call).
 * ------------------------------------------------------------------------------------------------ */
#define LC_SQNORM_BLOCKS 512 /* declared in (2i): the gradient may have any map element type */

/* ------------------------------------------------------------------------------------------------
 * (2g) Keypoint NLL of the sparse heads -- losses.py:318-326 sparse_kpt_loss: per-sample
 *      nll[b] = sum_{n,c} ( log std + |pts2d - project_apply(K, pts3d, R(q), t)| / std )   (the caller divides by B*N*2)
 *      and, when the pointers are non-NULL, d nll[b]/d pts2d and d nll[b]/d std (B,N,2) for a unit cotangent.
 * ------------------------------------------------------------------------------------------------ */
int lc_kpt_nll_fwd_bwd_f32(const float *K, const float *pose, const float *pts3d, const float *pts2d,
                           const float *pts2d_std, int B, int N, float *nll, float *d_pts2d, float *d_std, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2f) Test-time point selection of the dense heads (second half) -- test.py:39-45 quantile_msk,
 *      test.py:94-113 dense_point_select = mask | quantile | quantile_in_mask + the per-sample nonzero()/list/np.random
 *      padding, batched and compacted on the device.  Inputs are the dense front end's (B,N,.) rows (valid prefix
 *      in_counts[b], NULL = N; in_index (B,N) = source index of each entry for a second-stage selection, NULL = identity).
 *        mode 0: keep mask != 0                       (mask required: segmentation or RANSAC inliers)
 *        mode 1: keep w >= torch.quantile(w, q)       with w = inv_std[:,0] + inv_std[:,1]
 *        mode 2: q_b = 1 - (1-q) * mean(mask);  keep (w*mask >= torch.quantile(w*mask, q_b)) & mask
 *      Survivors are written in source order to the front of out_* (B,N,.); out_weights = inv_std, squared when
 *      square_weights (the inverse covariance lc_pnp_lm3_f32 takes with LC_PNP_WEIGHTS_ARE_ICOV -- pass 0 to keep inv_std);
 *      counts[b] = survivors, padded to min_count with seeded pseudo-random source indices when fewer survive
 *      (np.random.choice in the reference).  out_index may be NULL.
 * ------------------------------------------------------------------------------------------------ */
int lc_dense_select_f32(const float *pts2d, const float *inv_std, const float *pts3d, const unsigned char *mask,
                        const int *in_counts, const int *in_index, int B, int N, int mode, double quantile,
                        int square_weights, int min_count, unsigned seed, float *out_pts2d, float *out_weights,
                        float *out_pts3d, int *out_index, int *counts, void *stream);

/* The dense front end (2d) and the selection above in ONE launch for the test-time pipeline (declared in (2i) as lc_dense_frontend_select3),
 * one workgroup per object, for N = ceil((H-top)/sample) * ceil((W-left)/sample) <= 16384 sampled pixels (128x128 maps at stride 1,
 * configs/zlmo.yaml:30-37's test-time shape; more: an error, use the two launches).  The front end's (B,N,.) rows are never written; every selected value, count and index
 * equals what lc_dense_frontend_fwd3 followed by lc_dense_select_f32(mask = the visibility mask) returns, bit for bit
 * (tests/test_gpu_select.py).  xyz (B,3,H,W) required; vis_logits (B,H,W) required by modes 0 and 2. */
/* ------------------------------------------------------------------------------------------------
 * (2h) The dense heads' auxiliary losses of Loss_fn.forward (losses.py:281-316), one launch each way:
 *        losses[0] loss_noc        = mean |xyz * msk_noc - noc_tgt|                        (F.l1_loss, losses.py:293-295)
 *        losses[1] loss_seg        = mean seg(seg_logits, msk_vis)                         (losses.py:296)
 *        losses[2] loss_weight_seg = mean seg(wlogits, msk_vis broadcast over 2 channels)  (warm-up blend, losses.py:303-306)
 *      seg_type 0 = F.binary_cross_entropy_with_logits, 1 = Loss_seg_L1 (|sigmoid(x) - t|, losses.py:219-236).
 *      xyz (B,3,HW)|NULL with noc_tgt and the object mask as bool bytes (msk_noc_u8) or floats (msk_noc_f32), both (B,HW);
 *      seg_logits, msk_vis (B,HW); wlogits (B,2,HW)|NULL.  Forward: partials = 3 * 1024 doubles of workspace, ticket = LC_ARRIVAL_WORDS
 *      unsigned (the workgroups' arrival counters, sharded over 128-byte lines), zero before the first call (the kernel leaves them
 *      zero); sums in double precision, block partials added in block order.
 *      Backward: g_* = device scalars (the cotangents of the three means, NULL = none), d_* (same shapes as the inputs)|NULL.
 *      Declared in (2i) as lc_dense_aux_fwd2 / lc_dense_aux_bwd2.
 * ------------------------------------------------------------------------------------------------ */
#define LC_ARRIVAL_WORDS 544 /* (1 + 16 shards) x 32 words: lc_common.h kArrivalWords */
/* Loss_xyz_bin (losses.py:196-216), the ZebraPose heads' code loss: per-bit BCE-with-logits on logits * (msk_vis_logits > 0),
 * weighted by softmax(3 * min(h, 0.51 - h)) of the EMA histogram h of per-bit Hamming error rates inside that mask.  logits (B,C,HW),
 * gt_bits (B,C,HW) bool bytes, msk_vis_logits (B,HW), C <= 128.  Forward (one pass over the logits): histogram (C) is read and updated in
 * place (h <- h (1 - momentum) + rate momentum), loss (1), bin_weights (C) for the backward pass; partials = C * 32 * 3 doubles, ticket =
 * LC_ARRIVAL_WORDS unsigned, zero before the first call (left zero).  Backward: d_logits = g_loss (device scalar) * d loss / d logits.
 * Declared in (2i) as lc_xyz_bin_loss_fwd2 / lc_xyz_bin_loss_bwd2 (+ the counts / finish pair of a sharded batch). */
/* ------------------------------------------------------------------------------------------------
 * (2i) The f1 / f3 entry points and the Loss_fn glue for network outputs in any element type: fp32, or what a mixed-precision backbone emits
 *      (BASELINE.json configs[2] bf16, configs[4] fp16; ptnet.py:68-82 hands the heads' outputs over in the autocast type,
 *      losses.py:163-184,355-356 and floatbits.py:130-160,194-223 consume them).  map_dtype = LC_F32 | LC_F16 | LC_BF16 is the element
 *      type of EVERY `const void *` map argument and of every `void *` gradient map; everything else (per-sample scales, targets, masks,
 *      the (B,N,.) rows, cotangents of the rows, losses) stays fp32 / bytes.  Arithmetic is fp32 for every type and the element order of
 *      every reduction does not depend on it: a 16-bit map gives bit for bit the LC_F32 result on the up-cast values; a gradient map is that fp32
 *      gradient rounded to nearest even into the map's type.  No up-cast copy exists anywhere.  Four-element (8-byte) accesses need
 *      W (HW) % 4 == 0 and 8-byte aligned maps, else one element per access.
 *      wscale_dtype: the element type of the (B,) weight scale and of its gradient (fp32 under autocast, where exp is an fp32 op; the
 *      model's 16-bit type in a pure half-precision model).
 *      xyz_dtype (front end forward, front end + selection): the element type of `xyz` alone -- map_dtype, or LC_F32 next to 16-bit logits
 *      (test time with binary-code heads: the coordinate planes lc_bits_decode3 writes are fp32 whatever the network's type is).
 *      *_bstride: elements between consecutive samples of that input map, 0 = a dense batch.  The reference's heads are channel slices
 *      `out_raw[:, v]` of ONE (B,C_all,H,W) network output (ptnet.py:56): each sample's slice is contiguous, the batch stride is
 *      C_all*H*W -- passed here, the slice is consumed where it lies instead of being copied into shape first.  Gradient maps are dense.
 * ------------------------------------------------------------------------------------------------ */
int lc_dense_frontend_fwd3(const void *xyz, const void *wlogits, const void *wscale, const float *noc_scale,
                           const void *vis_logits, float vis_thresh, int map_dtype, int xyz_dtype, int wscale_dtype, long long xyz_bstride, long long wlogits_bstride, long long vis_bstride, int B, int H, int W, int top, int left,
                           int sample, float *pts2d, float *inv_std, float *pts3d, float *lse, unsigned char *vis_mask,
                           void *stream);
int lc_dense_frontend_bwd2(const void *wlogits, const void *wscale, const float *noc_scale, const float *lse,
                           const float *g_inv_std, const float *g_pts3d, int map_dtype, int wscale_dtype, long long wlogits_bstride, int B, int H, int W, int top, int left,
                           int sample, void *d_xyz, void *d_wlogits, void *d_wscale, void *stream);
/* Front end + selection (2f).  FEW objects with THOUSANDS of candidates each (zlmo's test-time shape: 64 objects x 16 384): given a workspace,
 * rows of more than 4096 candidates of at most 128 objects are selected by several workgroups per object -- each forms its share of the
 * log-sum-exp, of the radix select's histograms and of the compaction, the shares meet through ticketed words in the workspace -- instead of
 * one workgroup pulling the object's maps through one compute unit.  Every output bit for bit that of the one-workgroup form (workspace = NULL).
 *   lc_dense_frontend_select_workspace_bytes: bytes that shape needs (0: one workgroup per object anyway).
 *   workspace: that many bytes, 128-byte aligned, ZEROED ONCE by the caller, then owned by these calls (each leaves it ready for the next on the
 *       same stream).  NULL: one workgroup per object.  Scheduling as for lc_pnp_lm3_f32: the workgroups of an object wait for each other
 *       for a bounded time, and the call always enqueues the one-workgroup kernel behind them, which selects again -- bit for bit -- every
 *       object a workgroup gave up on and re-zeroes its region: contention costs time, never an object (tests/test_gpu_contention.py). */
size_t lc_dense_frontend_select_workspace_bytes(int B, int H, int W, int top, int left, int sample);
int lc_dense_frontend_select3(const void *xyz, const void *wlogits, const void *wscale, const float *noc_scale,
                              const void *vis_logits, float vis_thresh, int map_dtype, int xyz_dtype, int wscale_dtype, long long xyz_bstride, long long wlogits_bstride, long long vis_bstride, int B, int H, int W, int top, int left,
                              int sample, int mode, double quantile, int square_weights, int min_count, unsigned seed, int pose_index_offset,
                              float *out_pts2d, float *out_weights, float *out_pts3d, int *out_index, int *counts,
                              void *workspace, size_t workspace_bytes, void *stream);
int lc_bits_decode_gt_fwd3(const void *logits, const unsigned char *gt_bits, const unsigned char *gt_msk,
                           const float *out_scale, const float *out_xform, int map_dtype, long long logits_bstride, int B, int C, int H, int W, int n0,
                           int n1, int n2, int black_background, int top, int left, int sample, float *out, void *stream);
int lc_bits_decode_gt_bwd3(const void *logits, const unsigned char *gt_bits, const unsigned char *gt_msk,
                           const float *out_scale, const float *out_xform, const float *g_out, int map_dtype, long long logits_bstride, int B, int C,
                           int H, int W, int n0, int n1, int n2, int black_background, int top, int left, int sample,
                           void *d_logits, void *stream);
int lc_bits_decode3(const void *logits, const float *out_scale, const float *out_xform, int map_dtype, long long logits_bstride, int B, int C, int H,
                    int W, int n0, int n1, int n2, int black_background, int planar, float *out, void *stream);
/* Inference decode of the SELECTED pixels only (test time, test.py:67-136 with binary-code heads): entry k < rows_counts[b] of row b is sampled
 * pixel rows_index[b][k] of the (top, left, sample) grid -- what lc_dense_frontend_select3 (called with xyz = out_pts3d = NULL: the selection alone)
 * leaves in out_index / counts -- and its object coordinates `(noc * out_scale - T[:3,3]) @ T[:3,:3]` go to out_pts3d[b][k] (B,rows_N,3).  The floats
 * of lc_bits_decode3 at those pixels; a fifth of its work at zlmo's test-time shape. */
int lc_bits_decode_rows(const void *logits, const float *out_scale, const float *out_xform, int map_dtype, long long logits_bstride,
                        int B, int C, int H, int W, int n0, int n1, int n2, int black_background, int top, int left, int sample,
                        const int *rows_index, const int *rows_counts, int rows_N, float *out_pts3d, void *stream);
int lc_dense_aux_fwd2(const void *xyz, const unsigned char *msk_noc_u8, const float *msk_noc_f32, const float *noc_tgt,
                      const void *seg_logits, const float *msk_vis, const void *wlogits, int map_dtype, long long xyz_bstride, long long seg_bstride, long long wlogits_bstride, int B, int HW,
                      int seg_type, float *losses, double *partials, unsigned *ticket, void *stream);
int lc_dense_aux_bwd2(const void *xyz, const unsigned char *msk_noc_u8, const float *msk_noc_f32, const float *noc_tgt,
                      const void *seg_logits, const float *msk_vis, const void *wlogits, int map_dtype, long long xyz_bstride, long long seg_bstride, long long wlogits_bstride, int B, int HW,
                      int seg_type, const float *g_noc, const float *g_seg, const float *g_wseg, void *d_xyz, void *d_seg,
                      void *d_wlogits, void *stream);
int lc_xyz_bin_loss_fwd2(const void *logits, const unsigned char *gt_bits, const void *msk_vis_logits, int map_dtype, long long logits_bstride, long long vis_bstride, int B,
                         int C, int HW, float momentum, float *histogram, float *loss, float *bin_weights, double *partials,
                         unsigned *ticket, void *stream);
/* Loss_xyz_bin when the batch is SHARDED over ranks (collective 4): the reference is one process, so its histogram update
 * (losses.py:203-208) sees the whole batch's Hamming errors and visible pixels.  The one-launch forward above splits in two around the
 * caller's all-reduce, the same streaming pass and the same closing arithmetic (one device function), so a world of one gives the bits of
 * lc_xyz_bin_loss_fwd2:
 *   lc_xyz_bin_loss_counts: the pass over this rank's logits; counts (C + 1 64-bit integers, 8-byte aligned) <- per-bit errors inside the hard
 *     mask, then the mask's population; bce_mean (C) <- this rank's per-bit BCE means (stay on the device).  partials / ticket as above.
 *   -- the caller sums `counts` over the ranks (integers: exact at any batch size) --
 *   lc_xyz_bin_loss_finish: histogram (C, read and updated in place) from the summed counts, bin_weights (C), loss (1) = this rank's loss.
 * Backward: lc_xyz_bin_loss_bwd2 with those bin_weights. */
int lc_xyz_bin_loss_counts(const void *logits, const unsigned char *gt_bits, const void *msk_vis_logits, int map_dtype, long long logits_bstride,
                           long long vis_bstride, int B, int C, int HW, long long *counts, float *bce_mean, double *partials, unsigned *ticket,
                           void *stream);
int lc_xyz_bin_loss_finish(const long long *counts, const float *bce_mean, int C, float momentum, float *histogram, float *loss,
                           float *bin_weights, void *stream);
/* The NormClipper pair (2h) on a gradient of any map type: the hooks sit on the heads' outputs
 * (losses.py:343-352), so under mixed precision the gradient they clip is 16-bit; read and written in place of a cast each way. */
int lc_sqnorm(const void *x, int dtype, long long n, double *partials, unsigned *ticket, float *sq, int accumulate,
              const float *state, float *state_snapshot, void *stream);
int lc_norm_clip_apply(const void *grad, int dtype, long long n, const float *sq, const float *state_in, float initial_max_norm,
                       float scale, double momentum, void *out, float *state_out, float *norm_out, void *stream);
int lc_xyz_bin_loss_bwd2(const void *logits, const unsigned char *gt_bits, const void *msk_vis_logits,
                         const float *bin_weights, const float *g_loss, int map_dtype, long long logits_bstride, long long vis_bstride, int B, int C, int HW, void *d_logits,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LC_AMD_H */
