"""ORACLE (test infrastructure -- never imported by the product path).

CPU restatement in numpy of the reference's FFN + PR-GLS matching loop.  Each function cites
the reference lines it follows; the formulation (per-point neighbour loop, materialised pair
grid, dense m x n x 3 differences, LU solve) is kept in the reference's algorithmic shape so
that it also serves as the timed CPU baseline in bench.py.

PARITY STATUS.  Everything numpy/scipy/sklearn in the reference for this path is runnable in
the build container with the absent third-party modules stubbed at import, so these functions
are PINNED against golden vectors generated from the reference itself
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_match.py).
The one exception is ``ffn_forward``: the reference's FFN executes inside tensorflow==2.11
(not under /root/reference, not installed) -> "parity unpinned" for the dense-layer arithmetic;
it restates the Keras layer semantics of ffn.py:225-265 and is cross-checked against an
independent torch-CPU float64 evaluation.
"""
from __future__ import annotations

import numpy as np

LEAKY_ALPHA = 0.3
BN_EPS = 1e-3


# ----------------------------------------------------------------------------- normalisation
def normalize_points(points: np.ndarray, return_para: bool = False):
    """ffn.py:330-374: centre; divide by 3*std (ddof 0) of the projection on the first principal
    axis.  sklearn's PCA centres, takes the top right-singular vector; the sign is irrelevant."""
    if points.ndim != 2:
        raise ValueError(f"Points should be a 2D table, but get {points.ndim}D")
    if points.shape[1] != 3:
        raise ValueError(f"Points should have 3D coordinates, but get {points.shape[1]}D")
    mean = np.mean(points, axis=0)
    xc = points - mean
    _, _, vt = np.linalg.svd(xc, full_matrices=False)
    proj = xc @ vt[0]
    std = np.std(proj)
    norm = (points - mean) / (3 * std)
    return (norm, (mean, 3 * std)) if return_para else norm


# ----------------------------------------------------------------------------- kNN features
def knn_features(points: np.ndarray, k: int = 20) -> np.ndarray:
    """ffn.py:288-304 (== track.py:137-156): for every point the k+1 nearest neighbours
    (itself first, distance 0), mean of those k+1 distances, k relative coordinates divided by
    that mean, then the mean itself -> (N, 3k+1) float32."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    if n < k + 1:
        raise ValueError(f"Expected n_neighbors <= n_samples, but n_samples = {n}, n_neighbors = {k + 1}")
    out = np.zeros((n, 3 * k + 1), dtype=np.float32)
    for i in range(n):
        d = pts - pts[i]
        dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        order = np.lexsort((np.arange(n), dist))[:k + 1]
        mean_d = np.mean(dist[order])
        rel = (pts[order[1:]] - pts[order[0]]) / mean_d
        row = np.zeros(3 * k + 1)
        row[:3 * k] = rel.reshape(3 * k)
        row[3 * k] = mean_d
        out[i] = row
    return out


def pair_grid(feat_ref: np.ndarray, feat_tgt: np.ndarray) -> np.ndarray:
    """ffn.py:306-325: row t*n_ref + r = [features of ref r | features of tgt t]."""
    n, m = feat_ref.shape[0], feat_tgt.shape[0]
    left = np.broadcast_to(feat_ref[None, :, :], (m, n, feat_ref.shape[1])).reshape(m * n, -1)
    right = np.broadcast_to(feat_tgt[:, None, :], (m, n, feat_tgt.shape[1])).reshape(m * n, -1)
    return np.concatenate([left, right], axis=1)


# ----------------------------------------------------------------------------- FFN
def _bn(x, bn):
    dt = x.dtype
    inv = bn["gamma"].astype(dt) / np.sqrt(bn["var"].astype(dt) + dt.type(BN_EPS))
    return (x - bn["mean"].astype(dt)) * inv + bn["beta"].astype(dt)


def _leaky(x):
    return np.where(x >= 0, x, x * x.dtype.type(LEAKY_ALPHA))


def ffn_forward(w: dict, x, dtype=np.float32, batch: int = 65536) -> np.ndarray:
    """ffn.py:260-265.  x: (P,122) or [ (P,61), (P,61) ] (legacy two-input call, track.py:175).
    Dense(61->512,no bias)+BN+LeakyReLU shared by both halves; concat; Dense(1024->512,no
    bias)+BN+LeakyReLU; Dense(512->1)+sigmoid.  Returns (P,1)."""
    if isinstance(x, (list, tuple)):
        x = np.concatenate([np.asarray(x[0]), np.asarray(x[1])], axis=1)
    x = np.asarray(x, dtype=dtype)
    out = np.empty((x.shape[0], 1), dtype=dtype)
    w1 = w["w1"].astype(dtype); w2 = w["w2"].astype(dtype); w3 = w["w3"].astype(dtype); b3 = w["b3"].astype(dtype)
    for s in range(0, x.shape[0], batch):
        xb = x[s:s + batch]
        h1 = _leaky(_bn(xb[:, :61] @ w1, w["bn1"]))
        h2 = _leaky(_bn(xb[:, 61:] @ w1, w["bn1"]))
        h = _leaky(_bn(np.concatenate([h1, h2], axis=1) @ w2, w["bn2"]))
        out[s:s + batch] = 1.0 / (1.0 + np.exp(-(h @ w3 + b3)))
    return out


class FFNRef:
    """A `.predict`-compatible stand-in for the Keras FFN (used as the model object handed to
    the *reference's* initial_matching_* functions when generating golden vectors)."""

    def __init__(self, weights: dict, dtype=np.float32):
        self.w = weights
        self.dtype = dtype

    def predict(self, x, batch_size=None, **_):
        return ffn_forward(self.w, x, self.dtype)


def initial_matching(ffn_predict, ref: np.ndarray, tgt: np.ndarray, k: int = 20) -> np.ndarray:
    """ffn.py:268-327 / track.py:117-178 -> (m, n) float32 similarity matrix corr[t, r]."""
    fr = knn_features(ref, k)
    ft = knn_features(tgt, k)
    scores = ffn_predict(pair_grid(fr, ft))
    return np.reshape(scores, (tgt.shape[0], ref.shape[0]))


# ----------------------------------------------------------------------------- greedy matching
def simple_match(score_mxn: np.ndarray, threshold: float = 0.1):
    """trackerlite.py:242-259: repeatedly take the global maximum (first in row-major order on
    ties), record (ref, tgt), clear its row and column; stop below `threshold`."""
    work = np.array(score_mxn, copy=True)
    m, n = work.shape
    pairs = []
    for _ in range(n):
        flat = int(np.argmax(work))
        t, r = divmod(flat, n)
        if work[t, r] < threshold:
            break
        pairs.append((r, t))
        work[t, :] = 0
        work[:, r] = 0
    pairs = np.array(pairs)
    prior = np.full_like(work, 0.1 / (n - 1))
    for r, t in pairs:
        prior[t, r] = 0.9
    return prior, pairs


def legacy_prior(corr_mxn: np.ndarray, threshold: float = 0.5) -> np.ndarray:
    """track.py:58-70: rows start at 1/n; a matched row becomes 0.1/(n-1) with 0.9 at its pair."""
    work = np.array(corr_mxn, copy=True)
    m, n = work.shape
    prior = np.ones((m, n)) / n
    for _ in range(n):
        flat = int(np.argmax(work))
        t, r = divmod(flat, n)
        if work[t, r] < threshold:
            break
        prior[t, :] = 0.1 / (n - 1)
        prior[t, r] = 0.9
        work[t, :] = 0
        work[:, r] = 0
    return prior


# ----------------------------------------------------------------------------- PR-GLS (TrackerLite dialect)
def dist_squares(ref_nx3, tgt_mx3):
    """trackerlite.py:361-365 -> (m, n)."""
    d = ref_nx3[None, :, :] - tgt_mx3[:, None, :]
    return np.sum(np.square(d), axis=2)


def gaussian_kernel(ref_nx3, tgt_mx3, sigma_square):
    """trackerlite.py:368-372."""
    return np.exp(-dist_squares(ref_nx3, tgt_mx3) / (2 * sigma_square))


def estimate_posterior(prior_mxn, sigma_square, predicted_ref_nx3, tgt_mx3, ratio_outliers, vol=1):
    """trackerlite.py:375-382."""
    k = gaussian_kernel(predicted_ref_nx3, tgt_mx3, sigma_square)
    num = (1 - ratio_outliers) * prior_mxn * k / (2 * np.pi * sigma_square) ** 1.5
    den = np.sum(num, axis=1) + ratio_outliers / vol
    return num / den[:, None]


def solve_movements_ref(sigma_square, lambda_, posterior_mxn, ref_nx3, tgt_mx3, gram_nxn):
    """trackerlite.py:409-417: (G diag(colsum P) + lambda sigma^2 I)^T C^T = (Y^T P - X^T diag)^T."""
    n = ref_nx3.shape[0]
    colsum = np.sum(posterior_mxn, axis=0)
    coef = gram_nxn * colsum[None, :] + lambda_ * sigma_square * np.identity(n)
    dep = tgt_mx3.T @ posterior_mxn - ref_nx3.T * colsum[None, :]
    return np.linalg.solve(coef.T, dep.T).T


def prgls_with_two_ref(prior_mxn, tgt_mx3, ref_nx3, tracked_lx3, beta, lambda_, max_iteration=2000,
                       return_iters=False):
    """trackerlite.py:309-358.  Returns (moved tracked_lx3, posterior) [+ iteration count]."""
    gamma = 0.05
    gram_nn = gaussian_kernel(ref_nx3, ref_nx3, beta ** 2)
    gram_nl = gaussian_kernel(tracked_lx3, ref_nx3, beta ** 2)
    sigma2 = dist_squares(ref_nx3, tgt_mx3).mean() / 3
    pred_n = ref_nx3.copy()
    pred_l = tracked_lx3.copy()
    post = None
    it = 0
    for it in range(1, max_iteration):
        post = estimate_posterior(prior_mxn, sigma2, pred_n, tgt_mx3, gamma)
        c_3n = solve_movements_ref(sigma2, lambda_, post, pred_n, tgt_mx3, gram_nn)
        mov_n = (c_3n @ gram_nn).T
        mov_l = (c_3n @ gram_nl).T
        if it > 1:
            pred_n += mov_n
            pred_l += mov_l
        sp = np.sum(post)
        gamma = 1 - sp / tgt_mx3.shape[0]
        if gamma < 1e-4:
            gamma = 1e-4
        sigma2 = np.sum(dist_squares(pred_n, tgt_mx3) * post) / (3 * sp)
        if np.sqrt(np.sum(np.square(mov_n))) < 1e-3:
            break
    return (pred_l, post, it) if return_iters else (pred_l, post)


def prgls_quick(prior_mxn, tgt_mx3, tracked_nx3, beta, lambda_, max_iteration=2000, return_iters=False):
    """trackerlite.py:262-306 == prgls_with_two_ref with the tracked set equal to the ref set."""
    gamma = 0.05
    gram_nn = gaussian_kernel(tracked_nx3, tracked_nx3, beta ** 2)
    sigma2 = dist_squares(tracked_nx3, tgt_mx3).mean() / 3
    pred_n = tracked_nx3.copy()
    post = None
    it = 0
    for it in range(1, max_iteration):
        post = estimate_posterior(prior_mxn, sigma2, pred_n, tgt_mx3, gamma)
        c_3n = solve_movements_ref(sigma2, lambda_, post, pred_n, tgt_mx3, gram_nn)
        mov_n = (c_3n @ gram_nn).T
        if it > 1:
            pred_n += mov_n
        sp = np.sum(post)
        gamma = 1 - sp / tgt_mx3.shape[0]
        if gamma < 1e-4:
            gamma = 1e-4
        sigma2 = np.sum(dist_squares(pred_n, tgt_mx3) * post) / (3 * sp)
        if np.sqrt(np.sum(np.square(mov_n))) < 1e-3:
            break
    return (pred_n, post, it) if return_iters else (pred_n, post)


# ----------------------------------------------------------------------------- PR-GLS (legacy dialect)
def pr_gls_quick(X, Y, corr, BETA=300, max_iteration=20, LAMBDA=0.1, vol=1e8):
    """track.py:11-114: voxel units, gamma0 = 0.1, prior built inside with threshold 0.5, a fixed
    max_iteration-1 EM iterations, T_X recomputed from X each time, sigma^2 floored at 1.
    Returns (P, T_X, C)."""
    gamma = 0.1
    n = X.shape[0]
    m = Y.shape[0]
    gram = np.exp(-dist_squares(X, X) / (2 * BETA * BETA))
    C = np.zeros((3, n))
    sigma2 = np.sum(dist_squares(X, Y)) / (3 * n * m)
    prior = legacy_prior(corr, 0.5)
    T_X = X.copy()
    P = None
    for _ in range(1, max_iteration):
        p1 = prior * np.exp(-dist_squares(T_X, Y) / (2 * sigma2))
        den = np.sum(p1, axis=1) + gamma * (2 * np.pi * sigma2) ** 1.5 / ((1 - gamma) * vol)
        P = p1 / den[:, None]
        colsum = (np.ones((1, m)) @ P).reshape(n)
        a = gram * colsum[None, :] + LAMBDA * sigma2 * np.identity(n)
        b = Y.T @ P - X.T * colsum[None, :]
        C = np.linalg.solve(a.T, b.T).T
        T_X = (X.T + C @ gram).T
        mp = np.sum(P)
        gamma = 1 - mp / m
        sigma2 = np.sum(P * dist_squares(T_X, Y)) / (3 * mp)
        if sigma2 < 1:
            sigma2 = 1
    return P, T_X, C


def predict_one_rep(pred_pre_lx3, inter_nx3, beta, C_3xn):
    """tracker.py:1269-1289: X_pred + (C . exp(-|X_pred - X_inter|^2 / 2 beta^2))^T."""
    gram_nl = np.exp(-dist_squares(pred_pre_lx3, inter_nx3) / (2 * beta * beta))  # (n, l)
    return pred_pre_lx3 + (C_3xn @ gram_nl).T


def fit_ffn_prgls(ffn_predict, seg_pre_nx3, seg_tgt_mx3, beta, lambda_, max_iteration, rep=5, k=20):
    """tracker.py:1224-1267: rep x (FFN -> PR-GLS with beta*0.8^i), chaining the moved points."""
    inter = seg_pre_nx3.copy()
    C_t, beta_t, inter_t = [], [], []
    for i in range(rep):
        inter_t.append(inter)
        corr = initial_matching(ffn_predict, inter, seg_tgt_mx3, k)
        _, moved, C = pr_gls_quick(inter.copy(), seg_tgt_mx3, corr, BETA=beta * (0.8 ** i),
                                   max_iteration=max_iteration, LAMBDA=lambda_)
        inter = moved
        C_t.append(C)
        beta_t.append(beta * (0.8 ** i))
    return C_t, beta_t, inter_t


def predict_pos_once(ffn_predict, seg_src_nx3, tracked_src_lx3, seg_tgt_mx3, beta, lambda_, max_iteration,
                     rep=5, k=20):
    """tracker.py:1193-1222 (draw=False branch)."""
    C_t, beta_t, inter_t = fit_ffn_prgls(ffn_predict, seg_src_nx3, seg_tgt_mx3, beta, lambda_, max_iteration, rep, k)
    pred = tracked_src_lx3.copy()
    for C, b, inter in zip(C_t, beta_t, inter_t):
        pred = predict_one_rep(pred, inter, b, C)
    return pred


# ----------------------------------------------------------------------------- ensemble schedules
def get_volumes_list(current_vol, skip_volumes, sampling_number=20, adjacent=False, start_vol=1):
    """trackerlite.py:420-438."""
    assert current_vol > start_vol, f"current_vol (={current_vol}) should be larger than start_vol (={start_vol})"
    span = current_vol - start_vol
    if span < sampling_number:
        vols = list(range(start_vol, current_vol))
    elif adjacent:
        vols = list(range(current_vol - sampling_number, current_vol))
    else:
        interval = span // sampling_number
        start = span % sampling_number + start_vol
        vols = list(range(start, current_vol - interval + 1, interval))
    return [v for v in vols if v not in skip_volumes]


def get_reference_vols(ensemble, vol, adjacent=False):
    """track.py:575-610."""
    if not ensemble:
        return [vol - 1]
    if vol - 1 < ensemble:
        return list(range(1, vol))
    if adjacent:
        return list(range(vol - ensemble, vol))
    interval = (vol - 1) // ensemble
    start = (vol - 1) % ensemble + 1
    return list(range(start, vol - interval + 1, interval))


def trim_mean(stack_kxnx3: np.ndarray, cut: float = 0.1) -> np.ndarray:
    """scipy.stats.trim_mean(..., 0.1, axis=0) as used at trackerlite.py:123 / tracker.py:1508:
    sort along axis 0, drop int(cut*k) from each end, mean of the rest."""
    a = np.sort(np.asarray(stack_kxnx3), axis=0)
    k = a.shape[0]
    lo = int(cut * k)
    return a[lo:k - lo].mean(axis=0)
