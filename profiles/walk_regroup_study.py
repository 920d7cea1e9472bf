"""Host-side replay of the wave walk (one shared cursor, sleeping lanes) with subtrees handed over to private walks
where few lanes of the wave open them: counts, per threshold, the shared visits that remain, the lane-visits of the
private walks and the hand-overs, for a sample of waves.  Pure numpy (its own Morton sort and octree), no GPU.

usage: python3 profiles/walk_regroup_study.py [log2 bodies=18] [waves=60] [lanes=64]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.setrecursionlimit(10000)
from nbody_amd import physics                                               # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
nwaves = int(sys.argv[2]) if len(sys.argv) > 2 else 60
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else 64
THETA = 0.5
W = 65536.0

b = physics.plummer(1 << log2n, seed=1)
pos = b.pos.astype(np.float32).astype(np.float64)
mass = b.mass.astype(np.float32).astype(np.float64)
n = len(mass)
cell = np.floor((pos + W / 2) * (2 ** 21 / W)).astype(np.uint64)
key = np.zeros(n, dtype=np.uint64)
for bit in range(21):
    for ax in range(3):
        key |= ((cell[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
order = np.argsort(key, kind="stable")
key, pos, mass = key[order], pos[order], mass[order]
pm = np.concatenate([[0.0], np.cumsum(mass)])
pmx = np.concatenate([np.zeros((1, 3)), np.cumsum(mass[:, None] * pos, axis=0)])

com, nm, w2, skip = [], [], [], []


def emit(a, z, level):
    """bodies [a, z) share the first `level` digits; emits the compressed octree in pre-order"""
    if z - a == 1:
        com.append(pos[a]); nm.append(mass[a]); w2.append(0.0); skip.append(len(com))
        return
    while level < 21:                                  # skip one-child levels
        sh = np.uint64(3 * (20 - level))
        if (key[a] >> sh) != (key[z - 1] >> sh):
            break
        level += 1
    me = len(com)
    m = pm[z] - pm[a]
    com.append((pmx[z] - pmx[a]) / m); nm.append(m); w2.append((W / (1 << level) / THETA) ** 2); skip.append(0)
    if level >= 21:
        for i in range(a, z):
            emit(i, i + 1, 21)
    else:
        sh = np.uint64(3 * (20 - level))
        digits = ((key[a:z] >> sh) & np.uint64(7)).astype(np.int64)
        cuts = np.searchsorted(digits, np.arange(9))
        for d in range(8):
            if cuts[d + 1] > cuts[d]:
                emit(a + cuts[d], a + cuts[d + 1], level + 1)
    skip[me] = len(com)


emit(0, n, 0)
com = np.array(com); nm = np.array(nm); w2 = np.array(w2); skip = np.array(skip, dtype=np.int64)
total = len(com)
print(f"{n} bodies, {total} nodes", flush=True)


def private_visits(p, start, end):
    """node visits of one body's own walk over nodes [start, end)"""
    c, v = start, 0
    while c < end:
        v += 1
        d = com[c] - p
        if w2[c] > 0 and d @ d < w2[c]:
            c += 1
        else:
            c = skip[c]
    return v


THRESHOLDS = (0, 4, 8, 12, 16, 24, 32)
rng = np.random.default_rng(0)
starts = rng.choice(n // LANES, size=nwaves, replace=False) * LANES
res = {T: dict(shared=0, private=0, tasks=0, awake=0) for T in THRESHOLDS}
for T in THRESHOLDS:
    for s in starts:
        p = pos[s:s + LANES]
        my_skip = np.zeros(LANES, dtype=np.int64)
        cur = 0
        r = res[T]
        while cur < total:
            awake = my_skip <= cur
            d = com[cur] - p
            r2 = (d * d).sum(1)
            opens = awake & (r2 < w2[cur])
            takes = awake & ~opens
            r["shared"] += 1
            r["awake"] += int(awake.sum())
            k = int(opens.sum())
            if k and k <= T:
                # hand the subtree over to the lanes that open it; the wave passes over it
                for lane in np.nonzero(opens)[0]:
                    r["private"] += private_visits(p[lane], cur + 1, skip[cur])
                r["tasks"] += k
                my_skip[takes] = skip[cur]
                cur = skip[cur]
            elif k:
                my_skip[takes] = skip[cur]
                cur += 1
            else:
                cur = skip[cur]
    r = res[T]
    sh, pv, tk = r["shared"] / nwaves, r["private"] / nwaves, r["tasks"] / nwaves
    print(f"threshold {T:2d}: shared visits per wave {sh:8.0f} (awake lanes {r['awake'] / max(r['shared'], 1):5.1f}), private lane-visits per wave "
          f"{pv:8.0f}, hand-overs per wave {tk:7.0f} ({tk / LANES:5.1f} per body); cost in shared-visit units at 2 / 3 per dense private step: "
          f"{sh + 2 * pv / 64:8.0f} / {sh + 3 * pv / 64:8.0f}", flush=True)
