"""A numpy / pure-Python model of the device's BATCHED priority flood (csrc/ct_segment.hip, ws_flood_batch_kernel): several pops per round, chosen
so that the result is the sequential flood's.  Test infrastructure: it states the rule independently of the HIP code so that its exactness can
be checked against the oracle's sequential flood (oracle/watershed_ref.py::watershed) on the CPU.

The rule.  With LB = the smallest image value among the still unlabelled in-mask neighbours of ALL queue entries, everything the sequential
flood pushes from now on is one of those neighbours or a neighbour of something pushed later, i.e. has value >= LB.  So the queue entries with
value < LB (strictly) leave the heap in (value, age, index) order before anything pushed meanwhile.  A round pops them together (the top
entry always pops; at most `lanes` per round: a prefix of a valid batch is valid), a neighbour goes to the FIRST pop that touches it, pushes get
ages ordered by (pop order, direction order) -- the sequential flood's order of ages.  Entries without an unlabelled neighbour are dropped at
once (their pop labels and pushes nothing); bounds are refreshed lazily (an entry remembers which neighbour gave its bound); when more entries
qualify than the member table holds, the threshold moves half way towards the top entry's value until they fit."""
import numpy as np


def batched_flood(image, markers, mask, lanes=64, member_cap=256):
    """-> (labels, rounds, pops).  image: values to flood by (ascending), markers: int labels (0 = none), mask: bool; connectivity 1."""
    image = np.asarray(image, dtype=np.float64)
    shape = image.shape
    out = np.where(mask, markers, 0).astype(np.int32).ravel().copy()
    img = image.ravel()
    msk = np.asarray(mask, dtype=bool).ravel()
    strides = [int(np.prod(shape[a + 1:])) for a in range(len(shape))]
    offs = sorted([(-s, a, -1) for a, s in enumerate(strides)] + [(s, a, 1) for a, s in enumerate(strides)])   # ascending raveled offset

    def nbrs(i):
        ci = np.unravel_index(i, shape)
        for d, (off, ax, sgn) in enumerate(offs):
            c = ci[ax] + sgn
            if 0 <= c < shape[ax] and msk[i + off]:
                yield d, i + off

    def bound(i):
        """(smallest value among the unlabelled neighbours, that neighbour) or (None, None)"""
        best = (None, None)
        for _, j in nbrs(i):
            if out[j] == 0 and (best[0] is None or img[j] < best[0]):
                best = (img[j], j)
        return best

    # entry: [value, age, idx, bound value, bound neighbour]; seeds carry age 0 and fall through to the index
    queue = [[img[i], 0, int(i), None, -1] for i in np.flatnonzero(out)]
    base, rounds, pops = 1, 0, 0
    while queue:
        rounds += 1
        kept = []
        for e in queue:                                       # A: lazily refreshed bounds; dead entries leave
            if e[4] is not None and (e[4] < 0 or out[e[4]] != 0):
                e[3], e[4] = bound(e[2])
            if e[4] is not None:
                kept.append(e)
            else:
                pops += 1
        queue = kept
        if not queue:
            break
        key = lambda e: (e[0], e[1], e[2])
        top = min(queue, key=key)
        lb = min(e[3] for e in queue)
        members = [e for e in queue if e is top or e[0] < lb]
        if len(members) > member_cap:                         # the device's table of members is finite: any stricter threshold still selects a
            vals = np.sort(np.array([e[0] for e in members]))  # prefix of the pop order -- halve the distance to the top entry until they fit
            while len(members) > member_cap:
                nlb = top[0] + (lb - top[0]) / 2
                lb = top[0] if nlb == lb else nlb               # (one ulp apart: the midpoint rounds back)
                if int(np.searchsorted(vals, lb, "left")) + 1 <= member_cap or lb == top[0]:
                    members = [e for e in queue if e is top or e[0] < lb]
        members.sort(key=key)
        batch = members[:lanes]
        ids = {id(e) for e in batch}
        queue = [e for e in queue if id(e) not in ids]
        pops += len(batch)
        claims = {}
        for r, e in enumerate(batch):                         # D: a neighbour goes to the first pop that touches it
            for d, j in nbrs(e[2]):
                if out[j] == 0 and (j not in claims or r * 8 + d < claims[j][0]):
                    claims[j] = (r * 8 + d, out[e[2]])
        for j, (prio, lab) in claims.items():                 # E: label and push, ages in (pop order, direction) order
            out[j] = lab
            queue.append([img[j], base + prio, int(j), None, -1])
        base += lanes * 8
    return out.reshape(shape), rounds, pops
