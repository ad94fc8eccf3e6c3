"""Host-side mirror of the hot-path functions of the reference's ``CellTracker/track.py``.

    pr_gls_quick(X, Y, corr, BETA, max_iteration, LAMBDA, vol)   reference track.py:11-114
    initial_matching_quick(ffn_model, ref, tgt, k_ptrs)            reference track.py:117-178
    get_reference_vols / get_remote_vols                           reference track.py:575-610
"""
from __future__ import annotations

import numpy as np

from . import _dev
from .ffn import initial_matching_ffn


def pr_gls_quick(X, Y, corr, BETA=300, max_iteration=20, LAMBDA=0.1, vol=1E8):
    """Legacy PR-GLS in voxel units -> (P (m,n), T_X (n,3), C (3,n)), float64 numpy."""
    t = _dev.torch()
    X_d, Y_d = _dev.points_dev(X), _dev.points_dev(Y)
    corr_d = _dev.to_dev(np.asarray(corr, dtype=np.float32), t.float32)
    if tuple(corr_d.shape) != (Y_d.shape[0], X_d.shape[0]):
        raise ValueError(f"corr must have shape (len(Y), len(X)) = {(Y_d.shape[0], X_d.shape[0])}, got {tuple(corr_d.shape)}")
    P, TX, C = _dev.prgls_legacy(X_d, Y_d, corr_d, BETA, max_iteration, LAMBDA, vol)
    return P.cpu().numpy(), TX.cpu().numpy(), C.cpu().numpy()


def initial_matching_quick(ffn_model, ref, tgt, k_ptrs):
    """Same computation as ffn.initial_matching_ffn; a foreign model receives the two-input list."""
    return initial_matching_ffn(ffn_model, ref, tgt, k_ptrs, two_inputs=True)


def get_reference_vols(ensemble, vol, adjacent=False):
    """Source volumes of an ensemble prediction (reference track.py:575-599)."""
    if not ensemble:
        return [vol - 1]
    if vol - 1 < ensemble:
        return list(range(1, vol))
    if adjacent:
        return list(range(vol - ensemble, vol))
    return get_remote_vols(ensemble, vol)


def get_remote_vols(ensemble, vol):
    """Evenly spread previous volumes (reference track.py:602-610)."""
    interval = (vol - 1) // ensemble
    start = (vol - 1) % ensemble + 1
    return list(range(start, vol - interval + 1, interval))
