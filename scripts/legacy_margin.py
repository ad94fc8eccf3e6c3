"""Max error of the legacy-dialect PR-GLS against the reference's golden vectors (how much of the 1e-4 budget is used)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
track = importlib.import_module("3deecelltracker_amd.track")
g = np.load(os.path.join(ROOT, "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "match.npz"))
for n in (50, 113, 180):
    X, Y, corr = g[f"lg_X_{n}"], g[f"lg_Y_{n}"], g[f"lg_corr_{n}"]
    for tag, (beta, lam, mi) in {"a": (300, 0.1, 20), "b": (1000 * 0.8 ** 2, 1e-5, 10)}.items():
        P, TX, C = track.pr_gls_quick(X.copy(), Y, corr, BETA=beta, max_iteration=mi, LAMBDA=lam)
        print(f"n={n} {tag}: max|P err| {np.abs(P - g[f'lg_{tag}_P_{n}']).max():.2e}  max|TX err| {np.abs(TX - g[f'lg_{tag}_TX_{n}']).max():.2e} (budget 1e-4)")
