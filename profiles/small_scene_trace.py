"""The reference's default scene (1,002 bodies, theta 1) for 200 steps, for a kernel trace of what a step of a small
scene launches: rocprofv3 --kernel-trace --stats -- python3 profiles/small_scene_trace.py [collisions=0|1] [precision]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402

collide = len(sys.argv) > 1 and sys.argv[1] == "1"
precision = sys.argv[2] if len(sys.argv) > 2 else "f32"
b = physics.makebodies(1000, seed=1)
with Engine(mode="tree", precision=precision, theta=1.0, collisions=collide) as e:
    e.upload(b.pos, b.vel, b.mass)
    if collide:
        e.set_radius(b.radius)
    e.step(20)
    e.sync()
    t = time.perf_counter()
    e.step(200)
    e.sync()
    print(f"collisions {collide} {precision}: {(time.perf_counter() - t) * 5:.4f} ms per step", flush=True)
