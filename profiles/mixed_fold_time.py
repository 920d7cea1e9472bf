"""Time of the mixed-precision walk (NBODY_AMD_LIB picks the build): 1M Plummer sphere, both kernels; two 8M-body galaxies
(benchmark configuration 5's scene); a 2.1M-target rank of eight on the 16M galaxies is covered by bench.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402

print("library:", os.environ.get("NBODY_AMD_LIB", "in-tree"), flush=True)
for n, scene in ((1 << 20, "plummer"), (1 << 21, "plummer"), (1 << 24, "galaxies")):
    b = physics.plummer(n, seed=1) if scene == "plummer" else physics.galaxy_pair(n, seed=1)
    for knob in (("1", "2") if n <= 1 << 21 else (None,)):
        if knob is None:
            os.environ.pop("NBODY_WALK_BODIES_PER_LANE", None)
        else:
            os.environ["NBODY_WALK_BODIES_PER_LANE"] = knob
        with Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0 if scene == "galaxies" else 0.0) as e:
            e.upload(b.pos, b.vel, b.mass)
            e.step(3)
            e.sync()
            t = time.perf_counter()
            e.step(10)
            e.sync()
            ms = (time.perf_counter() - t) * 100
            st = e.stats()
            print(n, scene, "bodies per lane", knob, "step %.3f ms" % ms, "force kernel %.3f ms" % e.time_force_kernel(5),
                  "interactions", st.interactions, flush=True)
