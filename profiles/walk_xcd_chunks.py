"""Walk time against the way its workgroups are dealt to the XCDs: contiguous eighths (0) or chunks of 2^s workgroups in
turn (NBODY_WALK_XCD_CHUNK_LOG2 = s), on one engine.  fp32 Plummer spheres and the two-galaxy scene in mixed precision.
usage: python3 profiles/walk_xcd_chunks.py [log2 bodies ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nbody_amd import Engine, physics

for log2n in [int(a) for a in sys.argv[1:]] or [20, 21, 24]:
    b = physics.plummer(1 << log2n, seed=1)
    row = []
    for s in (0, 2, 4, 6, 8):
        os.environ["NBODY_WALK_XCD_CHUNK_LOG2"] = str(s)
        with Engine(mode="tree", precision="f32", theta=0.5) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(2)
            row.append(float(np.median([e.step_timed().ms_force for _ in range(5)])))
    print(f"plummer 2^{log2n} f32: walk ms for chunk log2 0/2/4/6/8:", [round(t, 3) for t in row], flush=True)
g = physics.galaxy_pair(1 << 23, seed=1)
row = []
for s in (0, 4, 6):
    os.environ["NBODY_WALK_XCD_CHUNK_LOG2"] = str(s)
    with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0) as e:
        e.upload(g.pos, g.vel, g.mass)
        e.step(2)
        row.append(float(np.median([e.step_timed().ms_force for _ in range(3)])))
print("galaxy pair 2x8M mixed: walk ms for chunk log2 0/4/6:", [round(t, 3) for t in row], flush=True)
