"""PR-GLS with a prior as a trained FFN would give it (true correspondences score high): iterations and time -- dev helper."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
_dev = importlib.import_module("3deecelltracker_amd._dev"); ffn_mod = importlib.import_module("3deecelltracker_amd.ffn")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(7)
ext = np.array([512.0, 512.0, 128.0])
x = rng.uniform(0, 1, (n, 3)) * ext
xn, (mean, scale) = ffn_mod.normalize_points(x, return_para=True)
a = np.eye(3) + (rng.uniform(0, 1, (3, 3)) - 0.5) * 0.2
yn = xn @ a + (rng.uniform(0, 1, xn.shape) - 0.5) * 0.004
rep = rng.choice(n, int(0.15 * n), replace=False)
yn[rep] = rng.uniform(-0.5, 0.5, (len(rep), 3))
perm = rng.permutation(n); yn = yn[perm]                      # target t = perm^-1 ... row t of y is x[perm[t]] moved
corr = rng.uniform(0, 0.05, (n, n)).astype(np.float32)        # [t, r]
keep = ~np.isin(perm, rep)
corr[np.arange(n)[keep], perm[keep]] = rng.uniform(0.7, 0.99, keep.sum()).astype(np.float32)
corr_d = torch.from_numpy(corr).cuda()
torch.cuda.synchronize(); tg = time.perf_counter()
pairs, npairs, prior_d = _dev.greedy_match(corr_d, 0.1, 0)
torch.cuda.synchronize(); print(f"greedy on sharp scores: {(time.perf_counter() - tg) * 1e3:.2f} ms, {int(npairs.item())} pairs")
xd, yd = _dev.points_dev(xn), _dev.points_dev(yn)
for _ in range(2): out = _dev.prgls_two_ref(prior_d, yd, xd, xd, 3.0, 3.0, 2000)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = _dev.prgls_two_ref(prior_d, yd, xd, xd, 3.0, 3.0, 2000)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
moved = out[0].cpu().numpy(); it = out[-1]
err = np.abs(moved[perm[keep]] - yn[keep]).max()
print(f"n={n}: PR-GLS {dt*1e3:.2f} ms, {it} iterations, max |moved - target| over true pairs {err:.2e} (normalised units)")
