"""How a launch of the tree walk fills and drains the GPU: begin and end of every wavefront, from a measurement build.

Build (on the CPU box): make -C nbody_amd/csrc timeline   (checks the device assembly like the product build, then writes
profiles/libnbody_timeline.so)
Run on the GPU box:  NBODY_AMD_LIB=$PWD/profiles/libnbody_timeline.so python3 profiles/walk_timeline.py [bodies_in_2^20 ...]

Prints, per scene: the span of the launch, the wavefronts' lifetimes, how many wavefronts were in flight over time, the
share of wave slots (8 per SIMD) that were occupied, and what list scheduling of the measured lifetimes on as many
slots gives in launch order and longest first (a model without contention: it bounds what reordering could buy)."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nbody_amd import Engine, physics  # noqa: E402
from nbody_amd.engine import load_library  # noqa: E402

SLOTS = 256 * 4 * 8
TICK_US = 0.01          # the constant clock runs at 100 MHz


def schedule(durations, slots):
    free = [0.0] * min(slots, len(durations))
    heapq.heapify(free)
    end = 0.0
    for d in durations:
        t = heapq.heappop(free) + d
        end = max(end, t)
        heapq.heappush(free, t)
    return end


def timeline(n, per_lane):
    os.environ["NBODY_WALK_BODIES_PER_LANE"] = str(per_lane)
    lib = load_library()
    lib.nbe_debug_walk_timeline.argtypes = [C.c_void_p, C.c_size_t]
    b = physics.plummer(n, seed=1)
    with Engine(mode="tree", precision="f32", theta=0.5) as e:
        e.upload(b.pos, b.vel, b.mass)
        e.step(2)
        e.sync()
        ms = e.time_force_kernel(3)
        e.step(1)
        e.sync()
        waves = (n + 64 * per_lane - 1) // (64 * per_lane)
        buf = np.zeros((waves, 4), dtype=np.uint64)
        assert lib.nbe_debug_walk_timeline(buf.ctypes.data_as(C.c_void_p), waves) == 0
    if os.environ.get("WALK_TIMELINE_DUMP"):
        np.save(os.path.join(os.environ["WALK_TIMELINE_DUMP"], f"timeline_{n}_{per_lane}.npy"), buf)
    t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64)
    keep = t1 > 0
    t0, t1 = t0[keep], t1[keep]
    origin = t0.min()
    b0 = (t0 - origin) * TICK_US; b1 = (t1 - origin) * TICK_US
    life = b1 - b0
    span = b1.max()
    print(f"{n} bodies, {per_lane} per lane: {len(life)} wavefronts, force kernel {ms:.3f} ms, span of the launch {span / 1e3:.3f} ms")
    print(f"  lifetimes us: mean {life.mean():.0f} std {life.std():.0f} min {life.min():.0f} p50 {np.median(life):.0f} p99 {np.quantile(life, .99):.0f} max {life.max():.0f}")
    marks = np.linspace(0, span, 21)[1:-1]
    inflight = [(int(((b0 <= m) & (b1 > m)).sum())) for m in marks]
    print("  in flight at 5 % steps of the span:", inflight)
    print(f"  occupied share of {SLOTS} wave slots over the span: {life.sum() / (span * SLOTS):.3f}")
    print(f"  last wavefront begins at {b0.max() / 1e3:.3f} ms; half of the slots are empty from {np.sort(b1)[-SLOTS // 2] / 1e3:.3f} ms on")
    order = np.argsort(b0, kind="stable")
    nat = schedule(life[order], SLOTS)
    lpt = schedule(np.sort(life)[::-1], SLOTS)
    print(f"  list scheduling of these lifetimes on {SLOTS} slots: launch order {nat / 1e3:.3f} ms, longest first {lpt / 1e3:.3f} ms, "
          f"sum / slots {life.sum() / SLOTS / 1e3:.3f} ms", flush=True)


sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 16]
for m in sizes:
    for per_lane in ((1, 2) if m <= 2 else (2,)):
        timeline(m << 20, per_lane)
