"""Host-side mirror of the reference's ``CellTracker/trackerlite.py`` hot path.

    TrackerLite(results_dir, ffn_model_name, proofed_coords_vol1, miss_frame, basedir)   :33-150
    simple_match, prgls_quick, prgls_with_two_ref                                        :242-358
    dist_squares, gaussian_kernel, estimate_posterior, solve_movements_ref              :361-417
    evenly_distributed_volumes, get_volumes_list                                         :420-438

numpy in / numpy out like the reference; the arithmetic runs in csrc/ct_match.hip.  Inside
TrackerLite.predict_cell_positions the whole chain (features -> FFN -> greedy prior -> PR-GLS) stays
on the device; only the final (l, 3) coordinates come back.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np

from . import _dev, _lib
from .coord_image_transformer import Coordinates
from .ffn import FFN, initial_matching_device, initial_matching_ffn, normalize_points

FIGURE = "figure"
COORDS_REAL = "coords_real"
LABELS = "labels"
TRACK_RESULTS = "track_results"
SEG = "seg"

BETA, LAMBDA, MAX_ITERATION = (3, 3, 2000)
K_POINTS = 20


# --------------------------------------------------------------------------------- numeric kernels
def simple_match(initial_match_matrix: np.ndarray, threshold=0.1):
    """reference :242-259 -> (normalized_prob (m,n) same dtype as the input, pairs_px2 (ref, tgt))."""
    t = _dev.torch()
    mat = np.asarray(initial_match_matrix)
    corr_d = _dev.to_dev(mat.astype(np.float32, copy=False), t.float32)
    pairs_d, np_d, prior_d = _dev.greedy_match(corr_d, threshold, 0)
    k = int(np_d.item())
    pairs = pairs_d[:k].cpu().numpy().astype(np.int64)
    if k == 0:
        pairs = np.array([])
    return prior_d.cpu().numpy().astype(mat.dtype, copy=False), pairs


def prgls_with_two_ref(init_match_mxn, ptrs_tgt_mx3, prts_ref_nx3, tracked_ref_lx3, beta: float, lambda_: float,
                       max_iteration: int = MAX_ITERATION) -> Tuple[np.ndarray, np.ndarray]:
    """reference :309-358 -> (moved tracked_ref_lx3, posterior_mxn)."""
    t = _dev.torch()
    prior_d = _dev.to_dev(np.asarray(init_match_mxn, dtype=np.float64), t.float64)
    out_l, _, post, _ = _dev.prgls_two_ref(prior_d, _dev.points_dev(ptrs_tgt_mx3), _dev.points_dev(prts_ref_nx3),
                                           _dev.points_dev(tracked_ref_lx3), beta, lambda_, max_iteration)
    return out_l.cpu().numpy(), post.cpu().numpy()


def prgls_quick(init_match_mxn, ptrs_tgt_mx3, tracked_ref_nx3, beta: float, lambda_: float,
                max_iteration: int = MAX_ITERATION) -> Tuple[np.ndarray, np.ndarray]:
    """reference :262-306 -> (moved tracked_ref_nx3, posterior_mxn)."""
    t = _dev.torch()
    prior_d = _dev.to_dev(np.asarray(init_match_mxn, dtype=np.float64), t.float64)
    _, out_n, post, _ = _dev.prgls_two_ref(prior_d, _dev.points_dev(ptrs_tgt_mx3), _dev.points_dev(tracked_ref_nx3), None,
                                           beta, lambda_, max_iteration, want_ref=True)
    return out_n.cpu().numpy(), post.cpu().numpy()


def _pairwise(fn_name, ref, tgt, *scalars):
    t = _dev.torch(); L = _lib.lib()
    r, g = _dev.points_dev(ref), _dev.points_dev(tgt)
    out = _dev.empty((g.shape[0], r.shape[0]), t.float64, r.device)
    _lib.check(getattr(L, fn_name)(r.data_ptr(), r.shape[0], g.data_ptr(), g.shape[0], *scalars, out.data_ptr(),
                                   _dev.stream(r.device)), fn_name)
    return out.cpu().numpy()


def dist_squares(ptrs_ref_nx3, ptrs_tgt_mx3) -> np.ndarray:
    """reference :361-365 -> (m, n)."""
    return _pairwise("ct_dist_squares", ptrs_ref_nx3, ptrs_tgt_mx3)


def gaussian_kernel(ptrs_ref_nx3, ptrs_tgt_mx3, sigma_square: float) -> np.ndarray:
    """reference :368-372 -> (m, n)."""
    return _pairwise("ct_gaussian_kernel", ptrs_ref_nx3, ptrs_tgt_mx3, float(sigma_square))


def estimate_posterior(prior_p_mxn, initial_sigma_square: float, predicted_ref_nx3, ptrs_tgt_mx3,
                       ratio_outliers: float, vol: float = 1) -> np.ndarray:
    """reference :375-382."""
    t = _dev.torch(); L = _lib.lib()
    prior = _dev.to_dev(np.asarray(prior_p_mxn, dtype=np.float64), t.float64)
    r, g = _dev.points_dev(predicted_ref_nx3), _dev.points_dev(ptrs_tgt_mx3)
    P = _dev.empty(tuple(prior.shape), t.float64, prior.device)
    _lib.check(L.ct_estimate_posterior(prior.data_ptr(), float(initial_sigma_square), r.data_ptr(), r.shape[0], g.data_ptr(),
                                       g.shape[0], float(ratio_outliers), float(vol), P.data_ptr(), _dev.stream(prior.device)),
               "ct_estimate_posterior")
    return P.cpu().numpy()


def softmax_normalize(similarity_matrix_mxn: np.ndarray) -> np.ndarray:
    """reference :385-386 (unused upstream; host one-liner)."""
    a = np.asarray(similarity_matrix_mxn)
    e = np.exp(a - a.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def row_wise_normalize(similarity_matrix_mxn: np.ndarray) -> np.ndarray:
    """reference :389-390"""
    a = np.asarray(similarity_matrix_mxn)
    return a / np.sum(a, axis=1, keepdims=True)


def non_max_suppression_normalize(similarity_matrix_mxn: np.ndarray, threshold=0.5) -> np.ndarray:
    """reference :393-406: the greedy one-to-one prior in its legacy form (1/n everywhere, matched rows 0.1/(n-1) with 0.9 at the
    pair) -- ct_greedy_match mode 1, the same kernel the legacy PR-GLS uses."""
    t = _dev.torch()
    mat = np.asarray(similarity_matrix_mxn)
    corr_d = _dev.to_dev(mat.astype(np.float32, copy=False), t.float32)
    _, _, prior_d = _dev.greedy_match(corr_d, threshold, 1)
    return prior_d.cpu().numpy().astype(mat.dtype, copy=False)


def solve_movements_ref(initial_sigma_square, lambda_, posterior_mxn, ptrs_ref_nx3, ptrs_tgt_mx3, scaling_factors_nxn):
    """reference :409-417 -> movements basis C (3, n).  `scaling_factors_nxn` is the (symmetric) Gram matrix."""
    t = _dev.torch(); L = _lib.lib()
    P = _dev.to_dev(np.asarray(posterior_mxn, dtype=np.float64), t.float64)
    G = _dev.to_dev(np.asarray(scaling_factors_nxn, dtype=np.float64), t.float64)
    r, g = _dev.points_dev(ptrs_ref_nx3), _dev.points_dev(ptrs_tgt_mx3)
    m, n = P.shape
    Cm = _dev.empty((3, n), t.float64, P.device)
    ws = _dev.workspace(L.ct_prgls_workspace_bytes(m, n, 0), P.device)
    _lib.check(L.ct_solve_movements(float(initial_sigma_square), float(lambda_), P.data_ptr(), r.data_ptr(), n, g.data_ptr(), m,
                                    G.data_ptr(), Cm.data_ptr(), ws.data_ptr(), ws.numel(), _dev.stream(P.device)),
               "ct_solve_movements")
    return Cm.cpu().numpy()


# --------------------------------------------------------------------------------- ensemble schedule
def evenly_distributed_volumes(current_vol: int, sampling_number: int, start_vol: int = 1) -> List[int]:
    """reference :420-424"""
    interval = (current_vol - start_vol) // sampling_number
    start = (current_vol - start_vol) % sampling_number + start_vol
    return list(range(start, current_vol - interval + 1, interval))


def get_volumes_list(current_vol: int, skip_volumes: List[int], sampling_number: int = 20, adjacent: bool = False,
                     start_vol: int = 1) -> List[int]:
    """reference :427-438"""
    assert current_vol > start_vol, f"current_vol (={current_vol}) should be larger than start_vol (={start_vol})"
    if current_vol - start_vol < sampling_number:
        vols_list = list(range(start_vol, current_vol))
    elif adjacent:
        vols_list = list(range(current_vol - sampling_number, current_vol))
    else:
        vols_list = evenly_distributed_volumes(current_vol, sampling_number, start_vol=start_vol)
    return [vol for vol in vols_list if vol not in skip_volumes]


# --------------------------------------------------------------------------------- device pipeline
def match_device(ffn_model: FFN, seg_t1_n, seg_t2_m, confirmed_l, beta, lambda_, max_iteration=MAX_ITERATION,
                 k=K_POINTS, threshold=0.1, prepared=None):
    """All-device TrackerLite step on *normalised* fp64 device points:
    FFN scores (seg_t1 vs seg_t2) -> greedy prior -> PR-GLS moving `confirmed_l` -> (l,3) device tensor.
    `prepared`: _dev.prgls_prepare_ref(seg_t1_n, beta) made ahead of the call (same results, the factorisation of seg_t1's Gram matrix is
    then not part of this chain)."""
    corr = initial_matching_device(ffn_model, seg_t1_n, seg_t2_m, k)
    _, _, prior = _dev.greedy_match(corr, threshold, 0)
    out_l, _, _, iters = _dev.prgls_two_ref(prior, seg_t2_m, seg_t1_n, confirmed_l, beta, lambda_, max_iteration,
                                            want_posterior=False, prepared=prepared)
    return out_l, iters


def match_device_batched(ffn_model: FFN, problems, beta, lambda_, max_iteration=MAX_ITERATION, k=K_POINTS, threshold=0.1):
    """match_device for a list of independent (seg_t1_n, seg_t2_m, confirmed_l) problems: FFN scores and the greedy priors of all
    problems in one chain of launches, then ALL PR-GLS runs in another (ct_prgls_two_ref_batched; the GPU retires the ~10
    tiny dependent kernels of an EM iteration at the same rate for one problem as for twenty).  The EM state is bit-identical to (and the moved tracked sets within ~1e-11 of)
    [match_device(...) for ...].  -> list of ((l, 3) device tensor, iterations)."""
    batch = []
    for seg_t1_n, seg_t2_m, confirmed_l in problems:
        _dev.check_match_sizes(seg_t1_n.shape[0], seg_t2_m.shape[0], k, "match_device_batched")
    if isinstance(ffn_model, FFN) and ffn_model._handle is not None and len(problems) > 1:
        # FFN scores and greedy priors of all problems as one chain of launches too (ct_match_front_batched)
        priors = _dev.match_front_batched(ffn_model._handle, [p[0] for p in problems], [p[1] for p in problems], k, threshold, 0)
        batch = [(priors[i], p[1], p[0], p[2]) for i, p in enumerate(problems)]
    else:
        for seg_t1_n, seg_t2_m, confirmed_l in problems:
            corr = initial_matching_device(ffn_model, seg_t1_n, seg_t2_m, k)
            _, _, prior = _dev.greedy_match(corr, threshold, 0)
            batch.append((prior, seg_t2_m, seg_t1_n, confirmed_l))
    res = _dev.prgls_two_ref_batched(batch, beta, lambda_, max_iteration)
    return [(r[0], r[3]) for r in res]


class TrackerLite:
    """Tracks cells from pre-computed segmentations with a trained FFN (reference :33-150)."""

    ensemble_batched = True  # the PR-GLS runs of one ensemble prediction share one chain of launches (False: one by one)

    def __init__(self, results_dir: str, ffn_model_name: str, proofed_coords_vol1: Coordinates,
                 miss_frame: List[int] = None, basedir: str = "ffn_models"):
        if miss_frame is not None and not isinstance(miss_frame, List):
            raise TypeError(f"miss_frame should be a list or None, but got {type(miss_frame)}")
        self.results_dir = Path(results_dir)
        (self.results_dir / TRACK_RESULTS / FIGURE).mkdir(parents=True, exist_ok=True)
        (self.results_dir / TRACK_RESULTS / COORDS_REAL).mkdir(parents=True, exist_ok=True)
        (self.results_dir / TRACK_RESULTS / LABELS).mkdir(parents=True, exist_ok=True)

        self.ffn_model = FFN()
        self.ffn_model_path = None
        for ext in (".h5", ".npz"):
            cand = Path(basedir) / (ffn_model_name + ext)
            if cand.exists() or ext == ".h5" and self.ffn_model_path is None:
                self.ffn_model_path = cand
                if cand.exists():
                    break
        try:
            self.ffn_model.load_weights(str(self.ffn_model_path))
        except (OSError, ValueError) as e:
            raise ValueError(f"Failed to load the FFN model from {self.ffn_model_path}: {e}") from e

        self.proofed_coords_vol1 = proofed_coords_vol1
        self.miss_frame = [] if miss_frame is None else miss_frame

    def predict_cell_positions(self, t1: int, t2: int, confirmed_coord_t1: Coordinates = None,
                               beta: float = BETA, lambda_: float = LAMBDA, draw_fig: bool = False):
        """reference :70-109"""
        assert t2 not in self.miss_frame
        segmented_pos_t1 = self._get_segmented_pos(t1)
        segmented_pos_t2 = self._get_segmented_pos(t2)
        if confirmed_coord_t1 is None:
            confirmed_coord_t1 = segmented_pos_t1

        # normalise the three point sets with the confirmed set's (mean, scale), match, de-normalise: all on the device
        confirmed_norm_d, para_d = _dev.normalize_points(_dev.points_dev(confirmed_coord_t1.real))
        seg_norm_t2_d, _ = _dev.normalize_points(_dev.points_dev(segmented_pos_t2.real), apply_para=para_d)
        seg_norm_t1_d, _ = _dev.normalize_points(_dev.points_dev(segmented_pos_t1.real), apply_para=para_d)
        tracked_norm_d, _ = match_device(self.ffn_model, seg_norm_t1_d, seg_norm_t2_d, confirmed_norm_d, beta, lambda_)
        tracked_coords_t2 = _dev.denormalize_points(tracked_norm_d, para_d).cpu().numpy()
        if draw_fig:
            raise NotImplementedError("figure drawing is outside the accelerated path")
        return Coordinates(tracked_coords_t2, interpolation_factor=self.proofed_coords_vol1.interpolation_factor,
                           voxel_size=self.proofed_coords_vol1.voxel_size, dtype="real")

    def predict_cell_positions_ensemble(self, skipped_volumes: List[int], t2: int, coord_t1: Coordinates,
                                        beta: float, lambda_: float, sampling_number: int = 20,
                                        adjacent: bool = False, t_start: int = 1):
        """reference :111-125.  The <= sampling_number (t1 -> t2) matches are independent: with
        torch.distributed initialised they are sharded over the ranks and gathered (parallel.py)."""
        from . import parallel
        t = _dev.torch()
        vols = get_volumes_list(current_vol=t2, skip_volumes=skipped_volumes, sampling_number=sampling_number,
                                adjacent=adjacent, start_vol=t_start)

        def members(t1_list):
            """This rank's share of the source volumes: every member normalised with its own confirmed set (reference :88-93),
            FFN + greedy one by one, all PR-GLS runs batched into one chain of launches, de-normalised; the Coordinates round
            trip of the reference (float32 raw storage, :103-105 then .real at :123) is reproduced on the way out."""
            assert t2 not in self.miss_frame
            seg2 = _dev.points_dev(self._get_segmented_pos(t2).real)
            probs, paras = [], []
            for t1 in t1_list:
                loaded = np.load(str(self.results_dir / TRACK_RESULTS / COORDS_REAL / f"coords{str(t1).zfill(6)}.npy"))
                c = Coordinates(loaded, coord_t1.interpolation_factor, coord_t1.voxel_size, dtype="real")
                conf_n, para = _dev.normalize_points(_dev.points_dev(c.real))
                s2, _ = _dev.normalize_points(seg2, apply_para=para)
                s1, _ = _dev.normalize_points(_dev.points_dev(self._get_segmented_pos(t1).real), apply_para=para)
                probs.append((s1, s2, conf_n)); paras.append(para)
            outs = match_device_batched(self.ffn_model, probs, beta, lambda_) if self.ensemble_batched else \
                [match_device(self.ffn_model, *p, beta, lambda_) for p in probs]
            res = []
            for (tracked_n, _), para in zip(outs, paras):
                real = _dev.denormalize_points(tracked_n, para).cpu().numpy()
                res.append(_dev.to_dev(Coordinates(real, self.proofed_coords_vol1.interpolation_factor,
                                                   self.proofed_coords_vol1.voxel_size, dtype="real").real, t.float64))
            return res
        stack = parallel.sharded_map_gather(None, vols, tail_shape=(coord_t1.cell_num, 3), dtype=t.float64,
                                            batch_fn=members)                                   # [k][l][3] fp64 device
        mean = _dev.trim_mean(stack, 0.1).cpu().numpy()
        return Coordinates(mean, interpolation_factor=self.proofed_coords_vol1.interpolation_factor,
                           voxel_size=self.proofed_coords_vol1.voxel_size, dtype="real")

    def match_by_ffn(self, t1: int, t2: int, confirmed_coord_t1: Coordinates = None):
        """reference :127-142 without the plot: returns pairs_px2 (ref, tgt)."""
        assert t2 not in self.miss_frame
        segmented_pos_t1 = self._get_segmented_pos(t1)
        segmented_pos_t2 = self._get_segmented_pos(t2)
        if confirmed_coord_t1 is None:
            confirmed_coord_t1 = segmented_pos_t1
        confirmed_norm_t1, (mean_t1, scale_t1) = normalize_points(confirmed_coord_t1.real, return_para=True)
        seg_norm_t2 = (segmented_pos_t2.real - mean_t1) / scale_t1
        matching_matrix = initial_matching_ffn(self.ffn_model, confirmed_norm_t1, seg_norm_t2, K_POINTS)
        _, pairs_px2 = simple_match(matching_matrix)
        return pairs_px2

    def _get_segmented_pos(self, t: int) -> Coordinates:
        """reference :144-150: seg/coords%06d.npy holds raw voxel coordinates."""
        return Coordinates(np.load(str(self.results_dir / SEG / f"coords{str(t).zfill(6)}.npy")),
                           interpolation_factor=self.proofed_coords_vol1.interpolation_factor,
                           voxel_size=self.proofed_coords_vol1.voxel_size, dtype="raw")
