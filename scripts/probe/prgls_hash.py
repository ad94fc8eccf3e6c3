"""Hash of whole PR-GLS runs (single + ragged batches, random and simple_match-style priors): run under two builds / switches to
check bit-identity (the same script is the child process of tests/test_gpu_match.py's bit-identity test)."""
import importlib, sys, hashlib
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
dev = importlib.import_module("3deecelltracker_amd._dev")
h = hashlib.sha256()
probs = []
for n, seed in ((50, 0), (113, 1), (301, 2), (600, 3), (599, 4)):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 3)) * 0.3
    m = n - (seed % 3)                                        # m != n, and sizes that are not multiples of 4
    b = (a[rng.permutation(n)] * 1.05 + rng.normal(size=(n, 3)) * 0.01)[:m]
    prior = torch.from_numpy(rng.uniform(0.0, 1.0, (m, n))).cuda()
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    trk = torch.from_numpy(a[: n - 7] + 0.001).cuda()         # tracked set != ref set, l != n
    probs.append((prior, tb, ta, trk))
    out = dev.prgls_two_ref(prior, tb, ta, trk, 3.0, 3.0, 60, want_posterior=True)
    for t in out[:-1]:
        if torch.is_tensor(t):
            h.update(t.cpu().numpy().tobytes())
    h.update(str(out[-1]).encode())
res = dev.prgls_two_ref_batched(probs, 3.0, 3.0, 60)          # ragged batch of 5: the row-group kernels
# priors as simple_match builds them (one value per row + at most one matched column; some rows unmatched), one problem of the batch
# with two odd entries in a row, which must send IT back to the dense read, and a large problem on the two-pass path (n > 1024)
sprobs = []
for n, seed in ((64, 5), (301, 6), (600, 7), (599, 8), (1100, 9)):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(n, 3)) * 0.3
    m = n - (seed % 3)
    perm = rng.permutation(n)
    b = (a[perm] * 1.05 + rng.normal(size=(n, 3)) * 0.01)[:m]
    pr = np.full((m, n), np.float32(0.1 / (n - 1)), dtype=np.float64)
    rows = np.flatnonzero(rng.uniform(size=m) < 0.8)
    pr[rows, perm[rows]] = np.float32(0.9)
    if seed == 8:
        pr[3, 5] = 0.25; pr[3, 9] = 0.125
    sprobs.append((torch.from_numpy(pr).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(a).cuda(), torch.from_numpy(a[: n - 7] + 0.001).cuda()))
res += dev.prgls_two_ref_batched(sprobs, 3.0, 3.0, 40)
torch.cuda.synchronize()
for r in res:
    for t in r[:3]:
        if torch.is_tensor(t):
            h.update(t.cpu().numpy().tobytes())
    h.update(str(r[3]).encode())
print("HASH", h.hexdigest())
