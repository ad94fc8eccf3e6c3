"""Device plumbing shared by the host-side mirrors: numpy <-> HBM (torch as allocator), the current
HIP stream, and thin typed wrappers over the C ABI for the matching kernels."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def torch():
    return _lib.require_gpu()


def stream(device=None):
    t = torch()
    return t.cuda.current_stream(device).cuda_stream


def to_dev(a, dtype, device=None):
    t = torch()
    if isinstance(a, t.Tensor):
        return a.to(device=device or "cuda", dtype=dtype).contiguous()
    arr = np.ascontiguousarray(np.asarray(a), dtype={t.float64: np.float64, t.float32: np.float32, t.int32: np.int32}[dtype])
    return t.from_numpy(arr).to(device or "cuda")


def points_dev(p, device=None):
    p = np.asarray(p) if not hasattr(p, "is_cuda") else p
    if p.ndim != 2 or p.shape[1] != 3:
        raise ValueError(f"expected an (n, 3) array of points, got {tuple(p.shape)}")
    return to_dev(p, torch().float64, device)


def empty(shape, dtype, device=None):
    t = torch()
    return t.empty(shape, dtype=dtype, device=device or "cuda")


def workspace(nbytes, device=None):
    return empty((int(nbytes),), torch().uint8, device)


# ----------------------------------------------------------------------------------------- size limits of the kernels
# (documented in include/ctamd.h; the reference's numpy code has none, so they are checked here with an actionable
# message before any kernel of the pipeline runs, instead of surfacing as CT_ESHAPE half-way through)
KNN_MAX_POINTS = 4096        # knn_features_kernel keeps one point set's distances per wave in LDS
KNN_MAX_K = 31
GREEDY_MAX_SIDE = 16384      # min(m, n): one bitonic sort of the accepted pairs in LDS
PRGLS_MAX_POINTS = 16384     # practical bound (the n x n fp64 Gram / posterior matrices: 2 GB each at 16384); no kernel limit
TRIM_MEAN_MAX_K = 64


def check_match_sizes(n_ref: int, n_tgt: int, k: int = 20, what: str = "matching"):
    if n_ref > KNN_MAX_POINTS or n_tgt > KNN_MAX_POINTS:
        raise ValueError(f"{what}: point sets of {n_ref} / {n_tgt} cells exceed the kNN-feature kernel's limit of "
                         f"{KNN_MAX_POINTS} points per set (include/ctamd.h, ct_knn_features)")
    if k > KNN_MAX_K:
        raise ValueError(f"{what}: k_ptrs = {k} exceeds the kNN-feature kernel's limit of {KNN_MAX_K} neighbours")
    if min(n_ref, n_tgt) > GREEDY_MAX_SIDE:
        raise ValueError(f"{what}: min(m, n) = {min(n_ref, n_tgt)} exceeds the greedy matcher's limit of {GREEDY_MAX_SIDE}")
    if n_ref > PRGLS_MAX_POINTS:
        raise ValueError(f"{what}: {n_ref} reference points exceed the PR-GLS dense M-step limit of {PRGLS_MAX_POINTS} "
                         f"(include/ctamd.h, ct_prgls_two_ref / ct_prgls_legacy)")


# ----------------------------------------------------------------------------------------- wrappers
def knn_features(points_d, k):
    t = torch(); L = _lib.lib()
    n = points_d.shape[0]
    if n < k + 1:
        raise ValueError(f"Expected n_neighbors <= n_samples,  but n_samples = {n}, n_neighbors = {k + 1}")
    if n > KNN_MAX_POINTS or k > KNN_MAX_K:
        raise ValueError(f"kNN features: n = {n} (limit {KNN_MAX_POINTS}) / k = {k} (limit {KNN_MAX_K}) unsupported by "
                         f"ct_knn_features")
    feat = empty((n, 3 * k + 1), t.float32, points_d.device)
    _lib.check(L.ct_knn_features(points_d.data_ptr(), n, k, feat.data_ptr(), stream(points_d.device)), "ct_knn_features")
    return feat


def greedy_match(corr_d, threshold, mode, want_prior=True):
    """corr_d fp32 [m][n] -> (pairs int32 [n][2] (ref, tgt), n_pairs tensor, prior fp64 [m][n] or None)."""
    t = torch(); L = _lib.lib()
    m, n = corr_d.shape
    if min(m, n) > GREEDY_MAX_SIDE:
        raise ValueError(f"greedy matching: min(m, n) = {min(m, n)} exceeds the kernel's limit of {GREEDY_MAX_SIDE}")
    pairs = empty((n, 2), t.int32, corr_d.device)
    npairs = empty((1,), t.int32, corr_d.device)
    prior = empty((m, n), t.float64, corr_d.device) if want_prior else None
    ws = workspace(L.ct_greedy_workspace_bytes(m, n), corr_d.device)
    _lib.check(L.ct_greedy_match(corr_d.data_ptr(), m, n, float(threshold), int(mode), pairs.data_ptr(), npairs.data_ptr(),
                                 prior.data_ptr() if prior is not None else None, ws.data_ptr(), ws.numel(),
                                 stream(corr_d.device)), "ct_greedy_match")
    return pairs, npairs, prior


class PreparedRef:
    """What prgls_two_ref needs of the reference set alone (ct_prgls_prepare_ref): made ahead of the match, e.g. on a second stream while
    the next volume's U-Net runs.  `event` marks its completion on the stream it was enqueued on."""

    def __init__(self, ref_d, beta, buf, event):
        self.ref_d, self.beta, self.buf, self.event = ref_d, float(beta), buf, event


def prgls_prepare_ref(ref_d, beta):
    """Enqueue (asynchronously, on the current stream) the Gram matrix of ref_d fp64 [n][3] and its low-rank factor -> PreparedRef."""
    t = torch(); L = _lib.lib()
    n = int(ref_d.shape[0])
    if n > PRGLS_MAX_POINTS:
        raise ValueError(f"PR-GLS: {n} reference points exceed the dense M-step limit of {PRGLS_MAX_POINTS}")
    buf = workspace(L.ct_prgls_prepared_bytes(n), ref_d.device)
    _lib.check(L.ct_prgls_prepare_ref(ref_d.data_ptr(), n, float(beta), buf.data_ptr(), buf.numel(), stream(ref_d.device)), "ct_prgls_prepare_ref")
    ev = t.cuda.Event(); ev.record(t.cuda.current_stream(ref_d.device))     # (the stream the call above was enqueued on, not the current DEVICE's)
    return PreparedRef(ref_d, beta, buf, ev)


def prgls_two_ref(prior_d, tgt_d, ref_d, tracked_d, beta, lambda_, max_iteration, want_posterior=True, want_ref=False, prepared=None):
    t = torch(); L = _lib.lib()
    m, n = prior_d.shape
    if n > PRGLS_MAX_POINTS:
        raise ValueError(f"PR-GLS: {n} reference points exceed the dense M-step limit of {PRGLS_MAX_POINTS}")
    l = 0 if tracked_d is None else tracked_d.shape[0]
    dev = prior_d.device
    out_l = empty((l, 3), t.float64, dev) if l else None
    out_n = empty((n, 3), t.float64, dev) if want_ref else None
    post = empty((m, n), t.float64, dev) if want_posterior else None
    ws = workspace(L.ct_prgls_workspace_bytes(m, n, l), dev)
    iters = C.c_int(0)
    args = (prior_d.data_ptr(), tgt_d.data_ptr(), m, ref_d.data_ptr(), n,
            tracked_d.data_ptr() if l else None, l, float(beta), float(lambda_), int(max_iteration),
            out_l.data_ptr() if l else None, out_n.data_ptr() if want_ref else None,
            post.data_ptr() if want_posterior else None, C.byref(iters), ws.data_ptr(), ws.numel())
    if prepared is not None:
        if prepared.ref_d is not ref_d or prepared.beta != float(beta):
            raise ValueError("prgls_two_ref: `prepared` was made for another reference set or beta")
        cur = t.cuda.current_stream(dev)
        cur.wait_event(prepared.event)                          # (made on another stream)
        prepared.buf.record_stream(cur); prepared.ref_d.record_stream(cur)
        _lib.check(L.ct_prgls_two_ref_prepared(*args, prepared.buf.data_ptr(), prepared.buf.numel(), stream(dev)), "ct_prgls_two_ref_prepared")
    else:
        _lib.check(L.ct_prgls_two_ref(*args, stream(dev)), "ct_prgls_two_ref")
    return out_l, out_n, post, iters.value


def prgls_two_ref_batched(problems, beta, lambda_, max_iteration, want_posterior=False, want_ref=False):
    """B independent prgls_with_two_ref problems in one chain of launches (ct_prgls_two_ref_batched).
    problems: list of (prior fp64 [m][n], tgt [m][3], ref [n][3], tracked [l][3] or None) device tensors (ragged sizes allowed).
    -> list of (out_tracked | None, out_ref | None, posterior | None, iterations).  The EM state (out_ref, posterior, iterations) is
    bit-identical to separate prgls_two_ref calls; out_tracked agrees to ~1e-11 (normalised units): the batched chain moves the tracked set
    once with the summed coefficients, the single call adds C_i G_ln every iteration (CT_DEFER_TRACKED=0 restores that form)."""
    t = torch(); L = _lib.lib()
    B = len(problems)
    if B == 0:
        return []
    dev = problems[0][0].device
    ms = [int(p[0].shape[0]) for p in problems]; ns = [int(p[0].shape[1]) for p in problems]
    ls = [0 if p[3] is None else int(p[3].shape[0]) for p in problems]
    if max(ns) > PRGLS_MAX_POINTS:
        raise ValueError(f"PR-GLS: {max(ns)} reference points exceed the dense M-step limit of {PRGLS_MAX_POINTS}")
    out_l = [empty((l, 3), t.float64, dev) if l else None for l in ls]
    out_n = [empty((n, 3), t.float64, dev) if want_ref else None for n in ns]
    post = [empty((m, n), t.float64, dev) if want_posterior else None for m, n in zip(ms, ns)]
    ptrs = lambda xs: (C.c_void_p * B)(*[None if x is None else x.data_ptr() for x in xs])
    im, in_, il = _lib.ivec(ms), _lib.ivec(ns), _lib.ivec(ls)
    ws = workspace(L.ct_prgls_batched_workspace_bytes(B, im, in_, il), dev)
    iters = (C.c_int * B)()
    _lib.check(L.ct_prgls_two_ref_batched(B, ptrs([p[0] for p in problems]), ptrs([p[1] for p in problems]), im,
                                          ptrs([p[2] for p in problems]), in_, ptrs([p[3] for p in problems]), il, float(beta),
                                          float(lambda_), int(max_iteration), ptrs(out_l), ptrs(out_n), ptrs(post), iters,
                                          ws.data_ptr(), ws.numel(), stream(dev)), "ct_prgls_two_ref_batched")
    return [(out_l[b], out_n[b], post[b], int(iters[b])) for b in range(B)]


def prgls_legacy(X_d, Y_d, corr_d, BETA, max_iteration, LAMBDA, vol, want_P=True):
    t = torch(); L = _lib.lib()
    n, m = X_d.shape[0], Y_d.shape[0]
    if n > PRGLS_MAX_POINTS:
        raise ValueError(f"PR-GLS (legacy): {n} reference points exceed the dense M-step limit of {PRGLS_MAX_POINTS}")
    dev = X_d.device
    P = empty((m, n), t.float64, dev) if want_P else None
    TX = empty((n, 3), t.float64, dev)
    Cm = empty((3, n), t.float64, dev)
    ws = workspace(L.ct_prgls_workspace_bytes(m, n, 0), dev)
    _lib.check(L.ct_prgls_legacy(X_d.data_ptr(), n, Y_d.data_ptr(), m, corr_d.data_ptr(), float(BETA), int(max_iteration),
                                 float(LAMBDA), float(vol), P.data_ptr() if want_P else None, TX.data_ptr(), Cm.data_ptr(),
                                 ws.data_ptr(), ws.numel(), stream(dev)), "ct_prgls_legacy")
    return P, TX, Cm


def gram_apply(pred_d, inter_d, C_d, beta):
    L = _lib.lib()
    _lib.check(L.ct_gram_apply(pred_d.data_ptr(), pred_d.shape[0], inter_d.data_ptr(), inter_d.shape[0], C_d.data_ptr(),
                               float(beta), stream(pred_d.device)), "ct_gram_apply")
    return pred_d


def legacy_predict_pos(ffn_handle, seg_pre_d, seg_tgt_d, tracked_pre_d, beta, lambda_, max_iteration, reps, k_ptrs=20, want_fit=False):
    """Tracker._predict_pos_once for one source volume as ONE native call (ct_legacy_predict_pos) -> pred [l][3]
    (+ C [reps][3][n], inter [reps][n][3] with want_fit).  The call releases the interpreter lock for the whole chain."""
    t = torch(); L = _lib.lib()
    n, m = seg_pre_d.shape[0], seg_tgt_d.shape[0]
    l = 0 if tracked_pre_d is None else tracked_pre_d.shape[0]
    dev = seg_pre_d.device
    pred = empty((l, 3), t.float64, dev) if l else None
    Cs = empty((reps, 3, n), t.float64, dev) if want_fit else None
    inter = empty((reps, n, 3), t.float64, dev) if want_fit else None
    ws = workspace(L.ct_legacy_predict_workspace_bytes(n, m, int(reps), int(k_ptrs)), dev)
    _lib.check(L.ct_legacy_predict_pos(ffn_handle, seg_pre_d.data_ptr(), n, seg_tgt_d.data_ptr(), m,
                                       tracked_pre_d.data_ptr() if l else None, l, float(beta), float(lambda_), int(max_iteration), int(reps),
                                       int(k_ptrs), pred.data_ptr() if l else None, Cs.data_ptr() if want_fit else None,
                                       inter.data_ptr() if want_fit else None, ws.data_ptr(), ws.numel(), stream(dev)), "ct_legacy_predict_pos")
    return (pred, Cs, inter) if want_fit else pred


def match_front_batched(ffn_handle, refs, tgts, k, threshold, mode=0):
    """FFN scores + greedy prior of B independent problems as ONE chain of launches (ct_match_front_batched)
    -> list of prior tensors fp64 [m_b][n_b]."""
    import ctypes as C
    t = torch(); L = _lib.lib()
    B = len(refs)
    ns = [int(x.shape[0]) for x in refs]; ms = [int(x.shape[0]) for x in tgts]
    dev = refs[0].device
    priors = [empty((ms[b], ns[b]), t.float64, dev) for b in range(B)]
    ws = workspace(L.ct_match_front_batched_workspace_bytes(B, max(ns), max(ms), int(k)), dev)
    rp = (C.c_void_p * B)(*[x.data_ptr() for x in refs]); tp = (C.c_void_p * B)(*[x.data_ptr() for x in tgts])
    pp = (C.c_void_p * B)(*[x.data_ptr() for x in priors])
    _lib.check(L.ct_match_front_batched(ffn_handle, B, rp, (C.c_int * B)(*ns), tp, (C.c_int * B)(*ms), int(k), float(threshold), int(mode), pp,
                                        ws.data_ptr(), ws.numel(), stream(dev)), "ct_match_front_batched")
    return priors


LEGACY_BATCH_MAX_POINTS = 132      # ct_legacy_predict_pos_batched: one workgroup solves a problem's dense M-step


def legacy_predict_pos_batched(ffn_handle, seg_pre_list, seg_tgt_d, tracked_pre_list, beta, lambda_, max_iteration, reps, k_ptrs=20):
    """B source volumes of one ensemble prediction as ONE chain of launches -> pred fp64 [B][l][3] (ct_legacy_predict_pos_batched)."""
    import ctypes as C
    t = torch(); L = _lib.lib()
    B = len(seg_pre_list)
    m, l = seg_tgt_d.shape[0], tracked_pre_list[0].shape[0]
    ns = [int(x.shape[0]) for x in seg_pre_list]
    dev = seg_tgt_d.device
    out = empty((B, l, 3), t.float64, dev)
    ws = workspace(L.ct_legacy_predict_batched_workspace_bytes(B, max(ns), m, l, int(reps), int(k_ptrs)), dev)
    pre = (C.c_void_p * B)(*[x.data_ptr() for x in seg_pre_list]); trk = (C.c_void_p * B)(*[x.data_ptr() for x in tracked_pre_list])
    nn = (C.c_int * B)(*ns)
    _lib.check(L.ct_legacy_predict_pos_batched(ffn_handle, B, pre, nn, seg_tgt_d.data_ptr(), m, trk, l, float(beta), float(lambda_),
                                               int(max_iteration), int(reps), int(k_ptrs), out.data_ptr(), ws.data_ptr(), ws.numel(),
                                               stream(dev)), "ct_legacy_predict_pos_batched")
    return out


def trim_mean(stack_d, cut=0.1):
    """stack_d fp64 [k][n][3] -> [n][3]  (scipy.stats.trim_mean(..., cut, axis=0))"""
    t = torch(); L = _lib.lib()
    k = stack_d.shape[0]
    if k > TRIM_MEAN_MAX_K:
        raise ValueError(f"trim_mean: {k} predictions exceed the kernel's limit of {TRIM_MEAN_MAX_K}")
    n3 = int(stack_d[0].numel())
    out = empty(tuple(stack_d.shape[1:]), t.float64, stack_d.device)
    _lib.check(L.ct_trim_mean(stack_d.contiguous().data_ptr(), k, n3, float(cut), out.data_ptr(), stream(stack_d.device)),
               "ct_trim_mean")
    return out


def normalize_points(points_d, apply_para=None):
    """points_d fp64 [n][3] device -> (normalised points, para [4] = mean xyz, scale) all on the device."""
    t = torch(); L = _lib.lib()
    n = points_d.shape[0]
    out = empty((n, 3), t.float64, points_d.device)
    para = apply_para if apply_para is not None else empty((4,), t.float64, points_d.device)
    _lib.check(L.ct_normalize_points(points_d.data_ptr(), n, apply_para.data_ptr() if apply_para is not None else None,
                                     out.data_ptr(), None if apply_para is not None else para.data_ptr(), stream(points_d.device)),
               "ct_normalize_points")
    return out, para


def denormalize_points(points_d, para):
    t = torch(); L = _lib.lib()
    out = empty(tuple(points_d.shape), t.float64, points_d.device)
    _lib.check(L.ct_denormalize_points(points_d.data_ptr(), points_d.shape[0], para.data_ptr(), out.data_ptr(), stream(points_d.device)),
               "ct_denormalize_points")
    return out
