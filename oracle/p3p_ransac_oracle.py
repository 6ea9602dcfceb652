"""CPU oracle of the GPU PnP initialiser (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Stands where `lib/pnp/cv2_solver.py:69-88` (cv2.solvePnPRansac, EPnP flag, 150 iterations) stands in the reference.
OpenCV's RNG and its EPnP minimal solver cannot be matched bit for bit (and OpenCV 4.6.0.66, pinned in
`scripts/req_0.txt:13`, is not in this image), so PARITY WITH OpenCV IS UNPINNED; what this oracle pins is the kernel's own
contract, with integer outputs compared exactly:

  * the hypothesis stream: hypothesis h of pose b draws its 4 point indices from the same counter-based hash of
    (seed, b, h) as `lc_pnp_init.hip` -- integer arithmetic, restated here bit for bit;
  * every hypothesis is solved by an INDEPENDENT float64 P3P (Grunert's elimination to a quartic in the depth ratio,
    roots by numpy's companion-matrix eigenvalues, rotation by orthogonal Procrustes/SVD) -- the kernel uses a different
    algorithm (pencil of quadrics, cubic, Gauss-Newton polish), so agreement is not shared code;
  * scoring, the (inlier count, inlier error, hypothesis id) arg-max and the winner's inlier index set follow the kernel's
    definition in float64; `decided` reports for which poses the float32 kernel MUST pick the same hypothesis: the winner's count lead
    exceeds the number of points that it and the rival have within `margin` of the inlier threshold, and no error-sum tie within
    `margin`; `mask_unsure` lists the points of the winner's inlier mask that float32 may decide either way (the rest is exact).

Round 5 -- the float32-faithful mode (`score_f32`, `ransac_f32`).  The kernel's scoring is division-free IEEE float32 (fused
multiply-adds and multiplies in a fixed order, lc_pnp_init.hip: inlier_q / chunk_error), so it can be restated operation by operation:
`fma32` below is an exactly rounded float32 fma built from float64 arithmetic, the error sums are added in the kernel's association
(even / odd points of a 64-point chunk, (even + odd) / tz^2 per chunk -- +inf when tz <= 0 --, chunks in order).  Given the float32 hypotheses (the kernel's own,
read back from its workspace -- a different P3P algorithm agrees with them to 1e-10, not to the bit), per-hypothesis counts, error sums,
the winner, its inlier count and its inlier mask are compared with EQUALITY for every pose; the float64 run above stays as the sanity
bound on the hypotheses themselves.
"""
from __future__ import annotations

import numpy as np

M32 = 0xFFFFFFFF


def hash_u32(a: int) -> int:
    """lowbias32, as `hash_u32` in lc_pnp_init.hip."""
    a &= M32
    a ^= a >> 16
    a = (a * 0x7FEB352D) & M32
    a ^= a >> 15
    a = (a * 0x846CA68B) & M32
    a ^= a >> 16
    return a


def sample_indices(seed: int, b: int, hyp: int, nl: int):
    """The 4 point indices of hypothesis `hyp` of pose `b` (3 for P3P + 1 to disambiguate), exactly as the kernel draws them."""
    h = hash_u32((seed & M32) ^ hash_u32((b * 0x9E3779B9 + hyp) & M32))
    idx = []
    for k in range(4):
        h = hash_u32((h + 0x6D2B79F5) & M32)
        v = h % nl
        for _ in range(8):
            if v not in idx:
                break
            v = (v + 1) % nl
        idx.append(v)
    return idx


def p3p_grunert(y, x):
    """All poses (R, t) with  s_i y_i = R x_i + t,  s_i > 0, for unit bearings y (3,3) and model points x (3,3) (rows)."""
    d12, d13, d23 = x[0] - x[1], x[0] - x[2], x[1] - x[2]
    c2, b2, a2 = d12 @ d12, d13 @ d13, d23 @ d23          # c = |x1-x2|, b = |x1-x3|, a = |x2-x3|
    n = np.cross(d12, d13)
    if not (n @ n > 1e-12 * c2 * b2):
        return []
    cg, cb, ca = y[0] @ y[1], y[0] @ y[2], y[1] @ y[2]     # gamma: (1,2), beta: (1,3), alpha: (2,3)
    # s2 = u s1, s3 = v s1.  (E1) b2 (1 + u^2 - 2 u cg) = c2 (1 + v^2 - 2 v cb);  (E2) b2 (u^2 + v^2 - 2 u v ca) = a2 (1 + v^2 - 2 v cb)
    # (E1) - (E2) is linear in u:  u * 2 b2 (v ca - cg) = (c2 - a2) (1 + v^2 - 2 v cb) - b2 (1 - v^2)  ->  u = N(v) / D(v)
    P = np.array([1.0, -2.0 * cb, 1.0])                    # 1 - 2 cb v + v^2   (highest power first)
    N = np.polysub((c2 - a2) * P, b2 * np.array([-1.0, 0.0, 1.0]))
    D = 2.0 * b2 * np.array([ca, -cg])
    # substitute into (E1) * D^2:  b2 (D^2 + N^2 - 2 cg N D) - c2 P D^2 = 0   -- a quartic in v
    D2 = np.polymul(D, D)
    quartic = np.polysub(b2 * np.polyadd(np.polyadd(D2, np.polymul(N, N)), -2.0 * cg * np.polymul(N, D)), c2 * np.polymul(P, D2))
    sols = []
    for v in np.roots(quartic):
        if abs(v.imag) > 1e-7 * max(1.0, abs(v.real)) or v.real <= 0:
            continue
        v = v.real
        den = np.polyval(D, v)
        if abs(den) < 1e-14:
            continue
        u = np.polyval(N, v) / den
        if u <= 0:
            continue
        s = np.array([1.0, u, v]) * np.sqrt(b2 / np.polyval(P, v))
        for _ in range(4):  # Newton on the three distance constraints (removes the root finder's 1e-10)
            r = np.array([s[0] ** 2 + s[1] ** 2 - 2 * cg * s[0] * s[1] - c2, s[0] ** 2 + s[2] ** 2 - 2 * cb * s[0] * s[2] - b2,
                          s[1] ** 2 + s[2] ** 2 - 2 * ca * s[1] * s[2] - a2])
            J = 2 * np.array([[s[0] - cg * s[1], s[1] - cg * s[0], 0.0], [s[0] - cb * s[2], 0.0, s[2] - cb * s[0]],
                              [0.0, s[1] - ca * s[2], s[2] - ca * s[1]]])
            try:
                s = s - np.linalg.solve(J, r)
            except np.linalg.LinAlgError:
                break
        if not (s > 0).all():
            continue
        z = s[:, None] * y
        # orthogonal Procrustes on the centred triangles (+ their normals, which fixes the reflection)
        zc, xc = z - z.mean(0), x - x.mean(0)
        A = np.vstack((zc, np.cross(zc[0] - zc[1], zc[0] - zc[2]))).T @ np.vstack((xc, np.cross(xc[0] - xc[1], xc[0] - xc[2])))
        U, _, Vt = np.linalg.svd(A)
        R = U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt
        t = z.mean(0) - R @ x.mean(0)
        if any(np.allclose(R, R2, atol=1e-7) and np.allclose(t, t2, atol=1e-6 * (1 + np.abs(t).max())) for R2, t2 in sols):
            continue  # double root
        sols.append((R, t))
    return sols


def ransac(K, pts3d, pts2d, count, reproj_err, iterations, seed, b, margin=1e-3):
    """One pose; every one of its `count` points is sampled from and scored, as `cv2.solvePnPRansac` does (`cv2_solver.py:72-75`).  Returns dict(invalid, best_hyp, n_inliers, inliers (sorted indices), R, t, decided, per_hyp_count)."""
    n = int(min(count, len(pts3d)))
    if n < 4:
        return dict(invalid=1, best_hyp=-1, n_inliers=0, inliers=np.zeros(0, np.int64), R=np.eye(3), t=np.zeros(3), decided=True)
    k = K.astype(np.float64).reshape(-1)
    idet = 1.0 / (k[0] * k[4] - k[1] * k[3])
    X = pts3d[:n].astype(np.float64)
    du, dv = pts2d[:n, 0].astype(np.float64) - k[2], pts2d[:n, 1].astype(np.float64) - k[5]
    # the kernel rounds the normalised image points to float32 once (the model points are float32 inputs)
    un = np.stack(((k[4] * du - k[1] * dv) * idet, (-k[3] * du + k[0] * dv) * idet), -1).astype(np.float32).astype(np.float64)
    Xs = X.astype(np.float32).astype(np.float64)
    nl = n
    thr = float(np.float32(reproj_err) * np.float32(np.sqrt(abs(idet))))
    thr2 = float(np.float32(thr) * np.float32(thr))
    rounds = (iterations + 63) // 64
    cand = []  # (count, err, hyp, R, t, uncertain)
    for hyp in range(rounds * 64):
        idx = sample_indices(seed, b, hyp, nl)
        yb = np.concatenate((un[idx[:3]], np.ones((3, 1))), 1)
        yb /= np.linalg.norm(yb, axis=1, keepdims=True)
        sols = p3p_grunert(yb, Xs[idx[:3]])
        pick, pick_e, pick_gap = None, np.inf, np.inf
        for R, t in sols:
            c = R @ Xs[idx[3]] + t
            if not c[2] > 0:
                continue
            e = float(((c[:2] / c[2] - un[idx[3]]) ** 2).sum())
            if e < pick_e:
                pick_gap = pick_e - e
                pick, pick_e = (R, t), e
            else:
                pick_gap = min(pick_gap, e - pick_e)
        if pick is None:
            cand.append((-1, np.inf, hyp, None, None, False, False))
            continue
        R, t = pick
        c = Xs[:nl] @ R.T + t
        # the kernel's division-free form (lc_pnp_init.hip: inlier_q): q = |c_xy - u cz|^2 < thr2 cz^2; inlier error = sum of q / tz^2
        e = ((c[:, :2] - un[:nl] * c[:, 2:3]) ** 2).sum(1)
        lim = thr2 * c[:, 2] ** 2
        inl = (c[:, 2] > 0) & (e < lim)
        # points the float32 kernel may legitimately put on the other side of the threshold (or of the camera plane): the hypothesis'
        # count is known up to that many
        n_unsure = int(((np.abs(e - lim) < margin * lim) | (np.abs(c[:, 2]) < 1e-6)).sum())
        ambiguous_pick = pick_gap < margin * max(pick_e, 1e-12)
        cand.append((int(inl.sum()), float(e[inl].sum() / t[2] ** 2) if t[2] > 0 else float(np.inf), hyp, R, t, n_unsure, ambiguous_pick))
    best = max(cand, key=lambda c: (c[0], -c[1], -c[2]))
    ok = best[0] >= 4
    out = dict(invalid=0 if ok else 1, best_hyp=best[2], best_count=best[0], per_hyp_count=np.array([c[0] for c in cand]))
    # decided: the float32 kernel cannot legitimately pick another hypothesis.  The winner's own 4th-point pick is unambiguous and its
    # count, less the points it has near the threshold, still beats every rival's count plus the rival's own near-threshold points
    # (plus 2 for a rival whose 4th-point pick is ambiguous: it may be another P3P root altogether); rivals that tie in an exactly
    # known count need a strict lead in the inlier error
    decided = not best[6] and (best[0] - best[5] >= 4 or not ok)
    for c in cand:
        if c[2] == best[2] or c[0] < 0:
            continue
        if c[0] + c[5] + (2 if c[6] else 0) < best[0] - best[5]:
            continue
        if c[5] == 0 and best[5] == 0 and not c[6] and c[0] == best[0]:
            decided = decided and (c[1] - best[1]) > margin * max(best[1], 1e-12)
        else:
            decided = False
    if not ok:
        out.update(n_inliers=0, inliers=np.zeros(0, np.int64), R=np.eye(3), t=np.zeros(3), decided=decided, mask_unsure=np.zeros(0, np.int64))
        return out
    R, t = best[3], best[4]
    Rf, tf = R.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64)
    c = X.astype(np.float32).astype(np.float64) @ Rf.T + tf
    un_all = np.stack(((k[4] * du - k[1] * dv) * idet, (-k[3] * du + k[0] * dv) * idet), -1).astype(np.float32).astype(np.float64)
    e = ((c[:, :2] - un_all * c[:, 2:3]) ** 2).sum(1)
    lim = thr2 * c[:, 2] ** 2
    inl = (c[:, 2] > 0) & (e < lim)
    # mask_unsure: the points of the winner's inlier mask float32 may decide either way; everywhere else the mask is exact
    mask_unsure = np.nonzero((np.abs(e - lim) < margin * lim) | (np.abs(c[:, 2]) < 1e-6))[0]
    out.update(n_inliers=int(inl.sum()), inliers=np.nonzero(inl)[0], R=R, t=t, decided=decided, mask_decided=decided and len(mask_unsure) == 0,
               mask_unsure=mask_unsure)
    return out


# ---- float32-faithful scoring (the kernel's arithmetic, operation by operation) ------------------------------------------------------

def fma32(a, b, c):
    """round-to-nearest-even float32 of a * b + c for float32 arrays, EXACTLY (one rounding): the product of two float32 is exact in
    float64; the float64 sum is rounded once, its rounding error recovered by TwoSum, and the second rounding (to float32) is corrected
    where the float64 sum sits exactly on a float32 midpoint and the discarded error decides the side."""
    a, b, c = (np.asarray(v, np.float32).astype(np.float64) for v in (a, b, c))
    p = a * b
    s = p + c
    with np.errstate(invalid="ignore"):
        bb = s - p
        err = (p - (s - bb)) + (c - bb)
    r = s.astype(np.float32)
    fin = np.isfinite(s) & np.isfinite(err)
    bits = np.ascontiguousarray(s).view(np.int64)
    mid = fin & ((bits & ((1 << 29) - 1)) == (1 << 28)) & (err != 0)
    if mid.any():
        r64 = r.astype(np.float64)
        up = np.where(r64 > s, r, np.nextafter(r, np.float32(np.inf)))
        down = np.where(r64 < s, r, np.nextafter(r, np.float32(-np.inf)))
        r = np.where(mid, np.where(err > 0, up, down), r)
    return r.astype(np.float32)


def normalised_points_f32(K, pts2d):
    """CamInv::normalise: K^-1 (u, v, 1) formed in double precision and rounded to float32 once (the subtraction inside is contracted to one
    fma by the build's -ffp-contract=on: the products are formed here in extended precision, where they are exact)."""
    k = np.asarray(K, np.float32).astype(np.float64).reshape(-1)
    idet = 1.0 / (k[0] * k[4] - k[1] * k[3])  # products of two floats are exact in double
    du = pts2d[:, 0].astype(np.float64) - k[2]
    dv = pts2d[:, 1].astype(np.float64) - k[5]
    ld = np.longdouble
    ux = ((ld(k[4]) * du.astype(ld) - (k[1] * dv).astype(ld)).astype(np.float64) * idet).astype(np.float32)
    uy = ((ld(-k[3]) * du.astype(ld) + (k[0] * dv).astype(ld)).astype(np.float64) * idet).astype(np.float32)
    return np.stack((ux, uy), -1), idet


def threshold2_f32(reproj_err_px, idet):
    thr = np.float32(reproj_err_px) * np.float32(np.sqrt(abs(idet)))
    return np.float32(thr * thr)


def inlier_q_f32(hyp32, X, un, thr2):
    """lc_pnp_init.hip inlier_q for hypotheses hyp32 (H,12) x points X (n,3), un (n,2), all float32 -> (inlier (H,n) bool, q (H,n) float32)."""
    R, t = hyp32[:, None, :9], hyp32[:, None, 9:]
    Xx, Xy, Xz = X[None, :, 0], X[None, :, 1], X[None, :, 2]
    cz = fma32(R[..., 8], Xz, fma32(R[..., 7], Xy, fma32(R[..., 6], Xx, t[..., 2])))
    cx = fma32(R[..., 2], Xz, fma32(R[..., 1], Xy, fma32(R[..., 0], Xx, t[..., 0])))
    cy = fma32(R[..., 5], Xz, fma32(R[..., 4], Xy, fma32(R[..., 3], Xx, t[..., 1])))
    with np.errstate(invalid="ignore", over="ignore"):
        rx, ry = fma32(-un[None, :, 0], cz, cx), fma32(-un[None, :, 1], cz, cy)
        q = fma32(ry, ry, (rx * rx).astype(np.float32))
        lim = ((np.float32(thr2) * cz).astype(np.float32) * cz).astype(np.float32)
        return (cz > 0) & (q < lim), q


def score_f32(hyp32, X, un, thr2):
    """Every hypothesis against every point of the pose, as the scoring kernels do it: -> (count (H,) int, error (H,) float32)."""
    hyp32, X, un = np.asarray(hyp32, np.float32), np.asarray(X, np.float32), np.asarray(un, np.float32)
    H, n = len(hyp32), len(X)
    inl, q = inlier_q_f32(hyp32, X, un, thr2)
    C = (n + 63) // 64
    qi = np.zeros((H, C * 64), np.float32)
    qi[:, :n] = np.where(inl, q, np.float32(0))
    qi = qi.reshape(H, C, 32, 2)
    even, odd = np.zeros((H, C), np.float32), np.zeros((H, C), np.float32)
    for k in range(32):  # sequential float32 adds, even and odd points apart
        even = (even + qi[:, :, k, 0]).astype(np.float32)
        odd = (odd + qi[:, :, k, 1]).astype(np.float32)
    tz = hyp32[:, 11]
    with np.errstate(divide="ignore", over="ignore"):
        scale = np.where(tz > 0, np.float32(1) / (tz * tz).astype(np.float32), np.float32(1)).astype(np.float32)
    chunk = ((even + odd).astype(np.float32) * scale[:, None]).astype(np.float32)
    chunk = np.where((tz > 0)[:, None], chunk, np.float32(np.inf))  # chunk_error: a hypothesis behind the camera ranks last among equal counts
    err = np.zeros(H, np.float32)
    for c in range(C):  # chunk order
        err = (err + chunk[:, c]).astype(np.float32)
    return inl.sum(1).astype(np.int64), err


def ransac_f32(K, pts3d, pts2d, count, reproj_err_px, hyp32):
    """The kernel's RANSAC decision for one pose given its float32 hypotheses (H,12) = [R row-major | t]: every integer output exactly.
    -> dict(invalid, best_hyp, n_inliers, inlier_mask (over the row), per_hyp_count, per_hyp_err)."""
    N = len(pts3d)
    n = int(min(count, N))
    if n < 4:
        return dict(invalid=1, best_hyp=-1, n_inliers=0, inlier_mask=np.zeros(N, bool), per_hyp_count=None, per_hyp_err=None)
    un, idet = normalised_points_f32(K, np.asarray(pts2d[:n], np.float32))
    thr2 = threshold2_f32(reproj_err_px, idet)
    X = np.asarray(pts3d[:n], np.float32)
    cnt, err = score_f32(hyp32, X, un, thr2)
    order = sorted(range(len(cnt)), key=lambda h: (-int(cnt[h]), float(err[h]), h))  # (count, -error, -id) arg-max: better_hyp
    win = order[0]
    ok = cnt[win] >= 4
    mask = np.zeros(N, bool)
    if ok:
        mask[:n] = inlier_q_f32(np.asarray(hyp32, np.float32)[win:win + 1], X, un, thr2)[0][0]
    return dict(invalid=0 if ok else 1, best_hyp=int(win) if ok else -1, n_inliers=int(mask.sum()), inlier_mask=mask, per_hyp_count=cnt, per_hyp_err=err)


def rot_to_quat(R):
    """wxyz, w >= 0 (same convention as `mat_to_quat` in the kernel)."""
    from scipy.spatial.transform import Rotation

    q = np.roll(Rotation.from_matrix(R).as_quat(), 1)
    return q if q[0] >= 0 else -q
