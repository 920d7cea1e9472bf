"""Prices the classic traversal (every lane walks alone, node records through vector loads) against the shared cursor:
time of one force pass of either on the same tree, node visits of the lanes, and whether the accelerations are the same
bits.  fp32 Plummer spheres, theta 0.5.
usage: python3 profiles/walk_private.py [log2 bodies ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbody_amd import Engine, physics

for log2n in [int(a) for a in sys.argv[1:]] or [20, 22]:
    b = physics.plummer(1 << log2n, seed=1)
    out = {}
    for knob in ("0", "1"):
        os.environ["NBODY_WALK_PRIVATE"] = knob
        os.environ["NBODY_WALK_BODIES_PER_LANE"] = "1"
        with Engine(mode="tree", precision="f32", theta=0.5) as e:
            e.upload(b.pos, b.vel, b.mass)
            ms = e.time_force_kernel(3)
            acc, _ = e.accel()
            out[knob] = (ms, acc, e.stats().interactions)
    same = np.array_equal(out["0"][1], out["1"][1])
    print(f"2^{log2n} bodies: shared cursor {out['0'][0]:.3f} ms, every lane alone {out['1'][0]:.3f} ms, force evaluations "
          f"{out['0'][2]} / {out['1'][2]}, same bits: {same}", flush=True)
