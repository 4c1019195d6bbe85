"""CPU: independent cross-checks of the oracle's restated third-party arithmetic with numpy / scipy —
the pieces no reference-held test pins: fitLine (Eigen SelfAdjointEigenSolver, geometry.cpp:42-59), fitPlane
(Eigen ColPivHouseholderQR, geometry.cpp:62-73), knnSearch (nanoflann, kdtree.cpp:10-28) and the robust
Levenberg-Marquardt solve (Ceres, registration-inl.h:30-56). What each bounds and to what tolerance is listed
in DESIGN.md section 2."""
import math

import numpy as np
import pytest
import scipy.linalg
import scipy.optimize
import scipy.spatial

import reference_kats as K


def _rng(seed):
    return np.random.default_rng(seed)


# ---- fitLine vs numpy.linalg.eigh ---------------------------------------------------------------------------------
@pytest.mark.parametrize("k", [3, 4, 5])
def test_fit_line_against_eigh(oracle, k):
    rng = _rng(10 + k)
    worst = 0.0
    for trial in range(200):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pts = rng.normal(size=3) * 5 + np.outer(rng.uniform(-0.5, 0.5, k), d) + rng.normal(size=(k, 3)) * 0.02
        a, b, cond = oracle.fit_line(pts)
        assert cond == np.finfo(float).max  # geometry.cpp:55-56: the condition number is never computed
        c = pts.mean(axis=0)
        w, V = np.linalg.eigh((pts - c).T @ (pts - c))
        dir_ = V[:, 2]
        assert np.allclose((a + b) / 2, c, atol=1e-12)                 # centre
        assert abs(np.linalg.norm(a - b) - 0.2) < 1e-12               # c +- 0.1 dir
        got = (a - b) / 0.2
        err = min(np.linalg.norm(got - dir_), np.linalg.norm(got + dir_))  # direction up to sign
        worst = max(worst, err)
        assert err < 1e-9 * max(1.0, w[2] / max(w[2] - w[1], 1e-300))  # conditioning of the eigenvector
    assert worst < 1e-7


# ---- fitPlane vs numpy.linalg.lstsq, and Eigen's rank-revealing semantics vs LAPACK's pivoted QR -------------------
@pytest.mark.parametrize("k", [4, 5])
def test_fit_plane_against_lstsq(oracle, k):
    rng = _rng(20 + k)
    for trial in range(200):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        basis = np.linalg.svd(n[None, :])[2][1:]              # two in-plane directions
        c = n * rng.uniform(1.0, 10.0)                         # (a plane through the origin is unrepresentable: geometry.h:113-115)
        pts = c + rng.uniform(-0.5, 0.5, (k, 2)) @ basis + rng.normal(size=(k, 1)) * 0.01 * n
        nrm, d, avg = oracle.fit_plane(pts)
        abc = np.linalg.lstsq(pts, np.ones(k), rcond=None)[0]  # P abc = 1 in the least-squares sense
        assert np.allclose(nrm, abc / np.linalg.norm(abc), atol=1e-9)
        assert abs(d - 1.0 / np.linalg.norm(abc)) < 1e-9 * max(1.0, d)
        assert abs(avg - np.mean(pts @ nrm - d)) < 1e-12      # signed mean


def _basic_solution(P, b, r):
    """solution of P x = b on the first r pivot columns of LAPACK's column-pivoted QR (dgeqp3: largest remaining
    column norm first, the rule Eigen's ColPivHouseholderQR uses too), zeros elsewhere — what
    ColPivHouseholderQR::solve returns when it counts r non-zero pivots (NOT the minimum-norm solution)"""
    Q, R, piv = scipy.linalg.qr(P, mode="economic", pivoting=True)
    y = scipy.linalg.solve_triangular(R[:r, :r], (Q.T @ b)[:r])
    x = np.zeros(P.shape[1])
    x[piv[:r]] = y
    return x


def test_fit_plane_rank_deficient_sets(oracle):
    """Collinear neighbour sets (a scan line seen edge-on): P abc = 1 has rank 2. Eigen's solve() drops a pivot only
    when its column norm is below (eps * max norm)^2 / rows * (rows - k) [RECALLED: m_nonzero_pivots, not rank()],
    which the rounding residue of the third column straddles: the answer is the basic solution on the two pivot
    columns when the pivot is dropped, otherwise some exact solution of the (consistent) system. Both are checked;
    the pseudo-inverse solution (numpy lstsq) is neither."""
    rng = _rng(31)
    basic = 0
    for trial in range(400):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        p0 = rng.normal(size=3) * 4 + 3 * d
        pts = p0 + np.outer(rng.uniform(-1, 1, 5), d)          # exactly one direction: rank 2 (p0 and d)
        x = _basic_solution(pts, np.ones(5), 2)
        nrm, dd, avg = oracle.fit_plane(pts)
        if np.allclose(nrm, x / np.linalg.norm(x), atol=1e-6) and abs(dd - 1.0 / np.linalg.norm(x)) < 1e-6 * max(1.0, dd):
            basic += 1
            mn = np.linalg.lstsq(pts, np.ones(5), rcond=None)[0]
            assert not np.allclose(nrm, mn / np.linalg.norm(mn), atol=1e-3) or abs(x[np.argmin(np.abs(x))]) < 1e-3
        else:
            assert np.abs(pts @ (nrm / dd) - 1.0).max() < 1e-6   # three pivots kept: still solves the system
    assert basic > 0.9 * 400


def test_fit_plane_near_degenerate_sets_are_full_rank(oracle):
    """nearly collinear sets (third singular value 1e-5 of the first): all three pivots count, the answer is the
    unique least-squares solution"""
    rng = _rng(32)
    for trial in range(200):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        p0 = rng.normal(size=3) * 4 + 3 * d
        pts = p0 + np.outer(rng.uniform(-1, 1, 5), d) + rng.normal(size=(5, 3)) * 1e-5
        nrm, dd, avg = oracle.fit_plane(pts)
        abc = np.linalg.lstsq(pts, np.ones(5), rcond=None)[0]
        cond = np.linalg.cond(pts)
        assert np.linalg.norm(nrm / dd - abc) < 1e-13 * cond * cond * np.linalg.norm(abc), trial


# ---- knnSearch vs scipy.spatial.cKDTree -------------------------------------------------------------------------------
@pytest.mark.parametrize("n,k,radius", [(5000, 5, 1.0), (5000, 5, -1.0), (300, 5, 2.0), (3, 5, 1.0), (20000, 8, 0.3)])
def test_knn_against_ckdtree(oracle, n, k, radius):
    rng = _rng(n + k)
    pts = rng.uniform(-5, 5, (n, 3))
    tree, ref = oracle.KDTree(pts), scipy.spatial.cKDTree(pts)
    for q in rng.uniform(-5.5, 5.5, (300, 3)):
        got = oracle.knn(tree, q, k, radius) if hasattr(oracle, "knn") else tree.knn(q, k, radius)
        dist, idx = ref.query(q, k=min(k, n))
        dist, idx = np.atleast_1d(dist), np.atleast_1d(idx)
        keep = np.isfinite(dist) & ((dist < radius) if radius > 0 else True)   # kdtree.cpp:25: strict, <= 0 disables
        assert list(got) == list(idx[keep])                                     # same set, ascending order
        assert list(got) == list(oracle.knn_bruteforce(pts, q, k, radius))
    assert len(oracle.KDTree(np.zeros((0, 3))).knn(np.zeros(3), 5, 1.0)) == 0   # empty tree (test_registration.cpp:177-199)


# ---- the robust LM solve vs scipy.optimize.least_squares(loss="huber") ----------------------------------------------------
def _rot(w):
    th = np.linalg.norm(w)
    if th < 1e-300:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * Kx @ Kx


def _residuals(x, moved_e, prim_e, moved_p, prim_p):
    """the reference's residual blocks as a function of a left increment (rotation vector, translation):
    registration-inl.h:92-117 + geometry-inl.h:21-33"""
    R, t = _rot(x[:3]), x[3:]
    pe, pp = moved_e @ R.T + t, moved_p @ R.T + t
    a, b = prim_e[:, :3], prim_e[:, 3:]
    re = np.linalg.norm(np.cross(pe - a, pe - b), axis=1) / np.linalg.norm(a - b, axis=1)
    rp = np.abs(np.einsum("ij,ij->i", prim_p[:, :3], pp) - prim_p[:, 3])
    return np.concatenate([re, rp])


def _huber_cost(r):
    s = r * r
    return 0.5 * np.sum(np.where(s <= 1.0, s, 2.0 * np.sqrt(s) - 1.0))  # Ceres HuberLoss(1.0): registration.cpp:56, :97


def _problem(oracle, T, outliers=0):
    tgt_e, tgt_p = K.registration_scene()
    src_e, src_p = K.transform_points(T, tgt_e), K.transform_points(T, tgt_p)
    if outliers:  # a few source points pushed off their planes: residuals beyond the Huber threshold
        src_p = src_p.copy()
        src_p[::max(1, len(src_p) // outliers)] += np.array([1.3, 1.4, 1.2])
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    ve, _, me, pe = oracle.associate(src_e, tgt_e, ident, False)
    vp, _, mp, pp = oracle.associate(src_p, tgt_p, ident, True)
    return (src_e, src_p, tgt_e, tgt_p), (me[ve], pe[ve], mp[vp], pp[vp])


def test_first_evaluation_cost_is_the_huber_cost(oracle):
    T = K.pose7(K.quat_angle_axis(0.02, (0.3, -0.5, 0.8) / np.linalg.norm((0.3, -0.5, 0.8))), (0.05, -0.03, 0.02))
    (se, sp, te, tp), data = _problem(oracle, T, outliers=40)
    prm = oracle.RegParams()
    prm.max_iterations = 1
    _, _, _, info = oracle.register_features(se, sp, te, tp, None, prm, want_info=True)
    r0 = _residuals(np.zeros(6), *data)
    assert (r0 > 1.0).sum() >= 10                                       # the outer (linear) region of the loss is exercised
    assert info[0].n_edge_assoc + info[0].n_plane_assoc == len(r0)
    assert abs(info[0].initial_cost - _huber_cost(r0)) < 1e-10 * _huber_cost(r0)


@pytest.mark.parametrize("outliers", [0, 40])
def test_lm_update_against_scipy_huber_least_squares(oracle, outliers):
    """one ICF iteration from a start close enough for the 4 LM iterations (registration-inl.h:52) to converge: the
    update must be the minimiser of the reference's robust objective, found here by an unrelated solver"""
    T = K.pose7(K.quat_angle_axis(2e-3, (0.3, -0.5, 0.8) / np.linalg.norm((0.3, -0.5, 0.8))), (2e-3, -1e-3, 1.5e-3))
    (se, sp, te, tp), data = _problem(oracle, T, outliers)
    prm = oracle.RegParams()
    prm.max_iterations = 1
    pose, _, _, info = oracle.register_features(se, sp, te, tp, None, prm, want_info=True)
    sol = scipy.optimize.least_squares(_residuals, np.zeros(6), args=data, loss="huber", f_scale=1.0, method="trf",
                                       xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale="jac")
    # scipy's cost is 0.5 * sum(rho(r^2)) with the same rho: the two objectives are the same function
    assert abs(sol.cost - _huber_cost(_residuals(sol.x, *data))) < 1e-9 * max(sol.cost, 1e-12)
    upd = np.array(list(info[0].update))
    R_or = _rot(2.0 * math.atan2(np.linalg.norm(upd[:3]), upd[3]) * upd[:3] / max(np.linalg.norm(upd[:3]), 1e-300))
    rot_diff = np.linalg.norm(R_or @ _rot(sol.x[:3]).T - np.eye(3))
    # Without outliers the cost goes to ~0 and Ceres' relative function tolerance (1e-6) never stops the solve early:
    # the 4 iterations reach the minimiser. With 40 residuals in the linear region of the loss the cost floor is ~40,
    # the function tolerance ends the solve one step short (its candidate is discarded), and the
    # update is the minimiser only to ~1e-5 — the level at which "path dependence"  lives.
    tol = 1e-6 if outliers == 0 else 1e-4
    assert rot_diff < tol and np.linalg.norm(upd[4:] - sol.x[3:]) < tol, (rot_diff, upd[4:] - sol.x[3:])
    assert info[0].final_cost <= _huber_cost(_residuals(np.zeros(6), *data))
    gap = info[0].final_cost - sol.cost
    assert -1e-9 * max(1.0, sol.cost) < gap < 2e-6 * max(1.0, sol.cost), gap  # within the function tolerance of the optimum
    print(f"outliers={outliers}: rotation diff {rot_diff:.2e}, translation diff {np.linalg.norm(upd[4:] - sol.x[3:]):.2e}, "
          f"cost gap {gap:.3e} of {sol.cost:.6g}, LM iterations {info[0].lm_iterations}")


# ---- the non-finite Jacobian path (geometry-inl.h:24-26 under autodiff, registration.h:171-173) -----------------------------
def nan_jacobian_scene():
    """Edge points exactly on their fitted lines: exactly representable lines along z (x = 1, y = 3 and x = -2,
    y = 0.5), registered against themselves from the identity. The point-to-line residual is then exactly 0 and its
    derivative 0/0. Planar points: the reference scene's planes, so that min_associations is met."""
    z = np.arange(-20, 21) * 0.0625
    edge = np.concatenate([np.stack([np.full_like(z, 1.0), np.full_like(z, 3.0), z], 1),
                           np.stack([np.full_like(z, -2.0), np.full_like(z, 0.5), z], 1)])
    return edge, K.registration_scene()[1]


def test_non_finite_jacobian_fails_the_solve_and_leaves_the_update_at_identity(oracle):
    edge, planar = nan_jacobian_scene()
    ident = np.array([0, 0, 0, 1.0, 0, 0, 0])
    v, _, moved, prim = oracle.associate(edge, edge, ident, False)
    a, b = prim[v][:, :3], prim[v][:, 3:]
    assert v.all() and (np.linalg.norm(np.cross(moved[v] - a, moved[v] - b), axis=1) == 0.0).all()  # exactly on the line
    pose, term, iters, info = oracle.register_features(edge, planar, edge, planar, None, None, want_info=True)
    # Ceres: "Initial residual and Jacobian evaluation failed" -> parameters untouched -> update = identity, which the
    # convergence test of registration-inl.h:68-73 then reads as converged
    assert np.array_equal(pose, ident) and term == oracle.CONVERGED and iters == 1
    assert np.array_equal(np.array(list(info[0].update)), ident) and info[0].lm_iterations == 0
    # a non-identity start that moves the points off the lines has finite Jacobians and is solved normally
    T = K.pose7(K.quat_angle_axis(0.01, (0, 1, 0)), (0.01, 0.0, 0.0))
    pose, term, iters = oracle.register_features(K.transform_points(T, edge), K.transform_points(T, planar), edge, planar)
    err = oracle.pose_compose(T, pose)
    assert oracle.quat_angular_distance(err[:4], [0, 0, 0, 1.0]) < 1e-4 and np.all(np.abs(err[4:]) < 1e-3)
