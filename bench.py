#!/usr/bin/env python3
"""Benchmark of the gravity engine's hot path (one step = force pass + fused kick-drift over all bodies).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run
with one rank per GPU.  W untimed steps, then exactly K timed steps bracketed by barrier + device
synchronisation, maximum over ranks, and rank 0 prints ONE JSON line on stdout (everything else goes to stderr).

Headline workload (config.workload = "direct_fp32_plummer_1M"): 1,048,576-body Plummer sphere, fp32 tiled
direct sum; value = body-body interactions per second (ordered pairs N*(N-1) per step) over the whole job.
With N GPUs the targets are sharded and positions/velocities all-gathered each step, total work fixed
("strong" scaling).  Extra, untimed-region results ride along in "extra": the Barnes-Hut path on the same
cloud (theta = 0.5) at 1M and at 16M bodies.

"roofline": dominant kernel of the headline = the direct-sum force kernel; bound = fp32 vector ALU
(157.3 TFLOP/s); achieved = interactions * 20 flop / mean kernel time, timed live with HIP events on the
engine's stream.  "cpu_baseline": the CPU oracle (a C port of the reference's loops; the Go reference cannot be
built here) on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_INTERACTION = 20          # 3 sub, 3 mul/fma (r^2), 1 rsqrt, 3 mul (m/r^3), 3 fma, rounded up as customary
FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X vector fp32 peak
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bodies", type=int, default=1 << 20, help="bodies of the headline workload")
    ap.add_argument("--workload", default="direct", choices=["direct", "tree"], help="path timed as the headline")
    ap.add_argument("--no-extra", action="store_true", help="skip the Barnes-Hut extras")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="bodies of the CPU direct-sum sample")
    return ap.parse_args()


def timed_steps(step_fn, sync_fn, barrier_fn, warmup, steps):
    for _ in range(warmup):
        step_fn()
    sync_fn(); barrier_fn(); sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn(); barrier_fn()
    return time.perf_counter() - t0


def traffic_from_profiles(workload, field="hbm_bytes_per_launch"):
    """A figure of the committed rocprofv3 --pmc passes (profiles/traffic.json, written by profiles/collect.sh):
    HBM bytes per launch of the dominant kernel by default; None if not collected for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get(field)
    except Exception:
        return None


def cpu_baseline_direct(sample):
    """The reference's direct path (serial i<j pair loop, main.go:107-128) as ported in oracle/nbody_oracle.c,
    one thread, on `sample` bodies drawn from the same Plummer cloud."""
    from nbody_amd import physics
    from oracle import oracle as O
    b = physics.plummer(sample, seed=1)
    t0 = time.perf_counter()
    _, _, st = O.forces(b.mass, b.pos, mode="direct")
    dt = time.perf_counter() - t0
    return {"value": st.interactions / dt, "unit": "interactions/s", "cores": 1, "kind": "port",
            "sample": f"direct pair loop of main.go:107-128 (C port, fp64, 1 thread as in the reference) on "
                      f"{sample} bodies of the same Plummer cloud, one force pass, {dt:.1f} s"}


def cpu_baseline_tree(n, theta):
    """The reference's tree path: serial octree build + 4 walk threads (main.go:133-149), C port, true-COM mode."""
    from nbody_amd import physics
    from oracle import oracle as O
    b = physics.plummer(n, seed=1)
    t0 = time.perf_counter()
    _, _, st = O.forces(b.mass, b.pos, mode="tree", com_mode="true", theta=theta, drop_tail=False, groups=4)
    dt = time.perf_counter() - t0
    return {"steps_per_s": 1.0 / dt, "interactions_per_s": st.interactions / dt, "cores": 4, "kind": "port",
            "build_s": st.build_seconds, "walk_s": st.walk_seconds,
            "sample": f"one force pass on {n} bodies, theta={theta}, serial build + 4 walk threads"}


def one_gpu_reference(key, field="steps_per_s"):
    """A figure of the newest committed one-GPU run of this bench (profiles/rNN_bench_default.json): what a run on
    several GPUs is compared with in its own line."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            v = d.get("extra", {}).get(key, {}).get(field) if key else d.get(field)
            if v is not None:
                return {"value": v, "from": os.path.basename(path)}
        except Exception:
            continue
    return None


def eight_ranks_on_one_gpu(b, one_gpu_ms, **engine_kw):
    """A scene on the key-range-distributed tree with eight ranks side by side on the current GPU.  Two figures per
    rank: the timed step (ten steps end to end, nothing drained in between, divided by the eight ranks that take turns on
    the GPU: kernels, launches, read-backs and host work of a rank, with device copies where a real run has its
    exchanges), and the phase table (phases drained one by one, medians over five steps)."""
    import torch
    from nbody_amd.disttree import DistRank, DistributedTree, LocalComm
    nr = 8
    m = len(b.mass)
    per = (m + nr - 1) // nr
    ranks = []
    try:
        for r in range(nr):
            dr = DistRank(r, nr, device=torch.cuda.current_device(), **engine_kw)
            ranks.append(dr)
            dr.upload(int(per * 1.25) + 4096, b.pos[r * per:(r + 1) * per], b.vel[r * per:(r + 1) * per], b.mass[r * per:(r + 1) * per], r * per)
            dr.use_torch_stream()
        tree = DistributedTree(ranks, LocalComm(nr))
        tree.distribute()
        tree.step(2)                    # (the first step moves every body to its owner)
        timed_s = timed_steps(lambda: tree.step(1), torch.cuda.synchronize, lambda: None, 1, 10)
        timed_ms = 1e3 * timed_s / 10 / nr
        runs = [tree.step_profiled() for _ in range(5)]
        keys = sorted({k for t in runs for k in t})      # (a run in which new splitters were chosen has a phase of its own)
        acc = {k: float(np.median([t.get(k, 0.0) for t in runs])) for k in keys}
        per_rank = {k: v / nr for k, v in acc.items()}
        return {"timed_ms_per_step_per_rank": timed_ms, "ratio_one_gpu_step_to_timed_rank_step": one_gpu_ms / timed_ms,
                "timed": "10 steps of all eight ranks end to end without drains, divided by 8",
                "walk_kernel": "two bodies per lane (a rank changes over at 1.5M targets in fp32, 1M in mixed precision)",
                "phase_ms_sum_over_ranks": acc, "phase_ms_per_rank": per_rank, "ms_per_step_per_rank": sum(per_rank.values()),
                "one_gpu_ms_per_step": one_gpu_ms, "ratio_one_gpu_step_to_rank_step": one_gpu_ms / sum(per_rank.values()),
                "imported_nodes_per_rank": tree.last_import_nodes, "migrated_bodies_per_rank": tree.last_migrated,
                "bodies_per_rank": [int(r.engine.stats().n_alive) for r in ranks], "rebalances": tree.rebalances}
    finally:
        for r in ranks:
            r.close()


def main():
    args = parse()
    # Libraries (RCCL prints a version banner) write to file descriptor 1; keep the real stdout for the one JSON
    # line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from nbody_amd import Engine, physics
    from nbody_amd.sharded import EngineShard, ShardedStepper

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # NBODY_BENCH_REHEARSE=gloo: every rank on GPU 0 and the collectives over gloo (RCCL refuses two ranks on one GPU), to
    # walk the code of an N-rank run on a one-GPU box; the figures of such a run mean nothing
    rehearse = os.environ.get("NBODY_BENCH_REHEARSE") == "gloo"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    red_dev = "cpu" if rehearse else "cuda"                        # where the small reductions of timings live
    force_dist = os.environ.get("NBODY_BENCH_FORCE_DIST") == "1"   # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        if force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    def sync():
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def run(mode, n, precision, theta, steps, warmup, seed=1, **engine_kw):
        b = physics.plummer(n, seed=seed)
        eng = Engine(mode=mode, precision=precision, theta=theta, device=local, **engine_kw)
        eng.upload(b.pos, b.vel, b.mass)
        del b
        if world > 1 or force_dist:
            # NBODY_BENCH_SPLIT=1: two ranges per rank, the first one's positions exchanged beside the second force pass
            stepper = ShardedStepper(EngineShard(eng, device_index=local), always_gather=force_dist,
                                     split=os.environ.get("NBODY_BENCH_SPLIT") == "1" and n % (512 * max(world, 1)) == 0)
            step_fn = stepper.step
        else:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            step_fn = eng.step
        sec = timed_steps(step_fn, sync, barrier, warmup, steps)
        if world > 1:
            t = torch.tensor([sec], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec = float(t.item())
        return eng, sec

    n = args.bodies
    headline_mode = args.workload
    eng, sec = run(headline_mode, n, "f32", 0.5, args.steps, args.warmup)
    # dominant kernel, timed with HIP events on the engine's stream (this rank's shard)
    kernel_ms = eng.time_force_kernel(3)
    st = eng.stats()
    if headline_mode == "direct":
        inter_per_step = n * (n - 1)
        kernel_inter = st.interactions
        workload = f"direct_fp32_plummer_{n >> 20}M" if n % (1 << 20) == 0 else f"direct_fp32_plummer_{n}"
    else:
        full = st.interactions
        if world > 1:
            t = torch.tensor([float(full)], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t)
            full = float(t.item())
        inter_per_step = full
        kernel_inter = st.interactions
        workload = f"bh_theta0.5_fp32_plummer_{n >> 20}M" if n % (1 << 20) == 0 else f"bh_theta0.5_fp32_plummer_{n}"
    value = inter_per_step * args.steps / sec
    achieved = kernel_inter * FLOP_PER_INTERACTION / (kernel_ms * 1e-3) / 1e12
    roofline = {"bound": "valu", "kernel": "direct_kernel_f32<float,4,64,FUSED,EQM> (equal masses: the mass is folded into G)" if headline_mode == "direct" else "walk_kernel<float>",
                "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                "traffic": traffic_from_profiles(workload), "kernel_ms": kernel_ms,
                "flop_per_interaction": FLOP_PER_INTERACTION,
                # clock the kernel held in the profiled run (GRBM_GUI_ACTIVE / 8 / kernel time); the peak assumes 2.4 GHz
                "profiled_clock_ghz": traffic_from_profiles(workload, "clock_ghz")}
    extra = {}
    if world > 1 or force_dist:
        extra["world"] = {"WORLD_SIZE": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                          "devices_visible": torch.cuda.device_count(), "local_rank": local}
        ref = one_gpu_reference(None, "steps_per_s")
        if ref:
            extra["world"]["headline_steps_per_s_over_one_gpu"] = {"ratio": (args.steps / sec) / ref["value"], "one_gpu": ref}
        # the headline step phase by phase (exchanges blocking, maximum over the ranks)
        prof = ShardedStepper(EngineShard(eng, device_index=local), always_gather=force_dist)
        prof.step_profiled()
        acc = {}
        for _ in range(2):
            for k, v in prof.step_profiled().items():
                acc[k] = acc.get(k, 0.0) + v / 2
        if world > 1:
            t = torch.tensor([acc[k] for k in sorted(acc)], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            acc = {k: float(v) for k, v in zip(sorted(acc), t.tolist())}
        extra["headline_sharded_phase_ms"] = acc
    eng.close()

    if not args.no_extra:
        other = "tree" if headline_mode == "direct" else "direct"
        m = 1 << 20
        e2, s2 = run(other, m, "f32", 0.5, max(args.steps, 10), max(args.warmup, 2))
        st2 = None
        if world == 1:
            st2 = e2.step_timed()
        kms = e2.time_force_kernel(3)
        s_inter = e2.stats().interactions
        if world > 1:
            t = torch.tensor([float(s_inter)], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t)
            s_inter = float(t.item())
        key = "bh_theta0.5_fp32_plummer_1M" if other == "tree" else "direct_fp32_plummer_1M"
        extra[key] = {"steps_per_s": max(args.steps, 10) / s2, "ms_per_step": 1e3 * s2 / max(args.steps, 10),
                      "interactions_per_step": s_inter, "interactions_per_s": s_inter * max(args.steps, 10) / s2,
                      "force_kernel_ms": kms,
                      "force_kernel_frac_of_fp32_valu_peak": (s_inter / world) * FLOP_PER_INTERACTION / (kms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
        if st2 is not None:
            nb = m
            extra[key].update({"ms_sort": st2.ms_sort, "ms_build": st2.ms_build, "ms_force": st2.ms_force,
                               "tree_nodes": int(st2.tree_nodes),
                               "sort_algorithmic_GBps": nb * 200.0 / (st2.ms_sort * 1e-3) / 1e9 if st2.ms_sort > 0 else None,
                               "build_algorithmic_GBps": nb * 130.0 / (st2.ms_build * 1e-3) / 1e9 if st2.ms_build > 0 else None,
                               # counter-based HBM bytes per step of the two phases (profiles/traffic.json), beside the
                               # algorithmic 200 and 130 bytes per body
                               "sort_traffic_bytes": traffic_from_profiles(key, "sort_hbm_bytes_per_step"),
                               "build_traffic_bytes": traffic_from_profiles(key, "build_hbm_bytes_per_step")})
        e2.close()
        # benchmark configuration 4: the 16M-body cloud, on every GPU count (steps/s at N GPUs over steps/s at 1 GPU
        # is the scaling figure it is judged by)
        m = 1 << 24
        e3, s3 = run("tree", m, "f32", 0.5, 5, 1)
        st3 = e3.step_timed() if world == 1 else None
        kms3 = e3.time_force_kernel(2)          # also counts this rank's force evaluations
        i3 = float(e3.stats().interactions)
        if world > 1:
            t = torch.tensor([i3], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t)
            i3 = float(t.item())
        extra["bh_theta0.5_fp32_plummer_16M"] = {"steps_per_s": 5 / s3, "ms_per_step": 1e3 * s3 / 5,
                                                 "interactions_per_step": i3, "interactions_per_s": i3 * 5 / s3, "force_kernel_ms": kms3,
                                                 "force_kernel": "walk_pair_kernel<float> (two bodies per lane)" if world == 1 else "by the shard's size",
                                                 "force_kernel_frac_of_fp32_valu_peak": (i3 / world) * FLOP_PER_INTERACTION / (kms3 * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
        if world > 1 or force_dist:
            ref = one_gpu_reference("bh_theta0.5_fp32_plummer_16M")
            if ref:
                extra["bh_theta0.5_fp32_plummer_16M"]["steps_per_s_over_one_gpu"] = {"ratio": (5 / s3) / ref["value"], "one_gpu": ref}
        if st3 is not None:
            extra["bh_theta0.5_fp32_plummer_16M"].update({
                "ms_sort": st3.ms_sort, "ms_build": st3.ms_build, "ms_force": st3.ms_force,
                "sort_algorithmic_GBps": m * 200.0 / (st3.ms_sort * 1e-3) / 1e9 if st3.ms_sort > 0 else None,
                "build_algorithmic_GBps": m * 130.0 / (st3.ms_build * 1e-3) / 1e9 if st3.ms_build > 0 else None})
        if world > 1 or force_dist:
            # where a sharded step spends its time (phases run one after the other, exchanges blocking; maximum over
            # the ranks): the replicated build, the own range's forces, the two all-gathers
            prof = ShardedStepper(EngineShard(e3, device_index=local), always_gather=force_dist)
            prof.step_profiled()
            acc = {}
            for _ in range(3):
                for k, v in prof.step_profiled().items():
                    acc[k] = acc.get(k, 0.0) + v / 3
            if world > 1:
                t = torch.tensor([acc[k] for k in sorted(acc)], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                acc = {k: float(v) for k, v in zip(sorted(acc), t.tolist())}
            extra["bh_theta0.5_fp32_plummer_16M"]["sharded_phase_ms"] = acc
        e3.close()
        # The same cloud and the same per-body walk with the key-range-distributed tree (nbe_dist_*): every body resident
        # on one GPU, every GPU sorts and builds its own N/G bodies, pruned subtrees are exchanged.
        # (one rank per GPU: measured last, under a watchdog, see distributed_tree_extra below)
        if world == 1 and not force_dist:
            b = physics.plummer(m, seed=1)
            extra["bh_theta0.5_fp32_plummer_16M_distributed_tree_8_ranks_on_one_gpu"] = eight_ranks_on_one_gpu(
                b, extra["bh_theta0.5_fp32_plummer_16M"]["ms_per_step"], precision="f32", theta=0.5)
            del b
        # the same cloud with the opening test taken per wavefront of 128 bodies (finer than the reference's per-body
        # test, not its interaction set; off by default), also on every GPU count
        e3w, s3w = run("tree", m, "f32", 0.5, 5, 1, per_wave=True)
        extra["bh_theta0.5_fp32_plummer_16M_per_wave_criterion"] = {"steps_per_s": 5 / s3w, "ms_per_step": 1e3 * s3w / 5}
        e3w.close()

    if not args.no_extra and (world > 1 or force_dist):
        # benchmark configuration 5 on every GPU count above one: the two 8,388,608-body galaxies in mixed precision
        # (fp32 pair terms, fp64 sums and integrator, eps = 1 m), targets sharded over the ranks, 100 fused steps with the
        # drift it is judged by (every rank holds all bodies, so rank 0 sums the energies of its own copy)
        gp = physics.galaxy_pair(1 << 23, seed=1)
        e5 = Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, device=local)
        e5.upload(gp.pos, gp.vel, gp.mass)
        pscale = float((gp.mass * np.linalg.norm(gp.vel, axis=1)).sum())
        del gp
        ke0, pe0, p0 = e5.energy()
        st5 = ShardedStepper(EngineShard(e5, device_index=local), always_gather=force_dist)
        sec5 = timed_steps(st5.step, sync, barrier, 2, 98)
        if world > 1:
            t = torch.tensor([sec5], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sec5 = float(t.item())
        sync()
        ke1, pe1, p1 = e5.energy()
        extra["galaxy_pair_2x8M_bh_mixed_eps1_sharded"] = {
            "bodies": 1 << 24, "ms_per_step": 1e3 * sec5 / 98, "steps_per_s": 98 / sec5, "steps": 100,
            "bodies_left": int(e5.stats().n_alive),
            "drift": {"after_100_steps": {"energy_drift_rel": abs((ke1 + pe1) - (ke0 + pe0)) / abs(ke0 + pe0),
                                          "momentum_drift_rel": float(np.linalg.norm(p1 - p0)) / pscale}},
            "stated_bound": "|dE/E0| <= 2e-3 and |dP|/sum m|v| <= 1e-5 after 100 steps (the bounds tests/test_gpu_config5.py asserts on one GPU)",
            "parallelism": f"targets sharded over {world} GPU(s), replicated build, all-gather of pos+vel per step"}
        ref = one_gpu_reference("galaxy_pair_2x8M_bh_mixed_eps1")
        if ref:
            extra["galaxy_pair_2x8M_bh_mixed_eps1_sharded"]["steps_per_s_over_one_gpu"] = {"ratio": (98 / sec5) / ref["value"], "one_gpu": ref}
        e5.close()

    if not args.no_extra and world == 1:
        # benchmark configuration 2: 65,536-body uniform cloud, fp32 direct sum
        u = physics.uniform_cube(65536, seed=1)
        e4 = Engine(mode="direct", precision="f32", device=local)
        e4.upload(u.pos, u.vel, u.mass)
        e4.set_stream(torch.cuda.current_stream().cuda_stream)
        s4 = timed_steps(lambda: e4.step(1), sync, barrier, 5, 50)
        kms4 = e4.time_force_kernel(5)
        extra["direct_fp32_uniform_64k"] = {"ms_per_step": 1e3 * s4 / 50, "interactions_per_s": 65536 * 65535 * 50 / s4,
                                            "force_kernel_ms": kms4,
                                            "force_kernel_frac_of_fp32_valu_peak": 65536 * 65535 * FLOP_PER_INTERACTION / (kms4 * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
        e4.close()
        # the 1M-body Barnes-Hut step with the opening test taken per wave of 128 bodies instead of per body: finer
        # interactions than the reference's criterion asks for, not its interaction set (off by default)
        pw = physics.plummer(1 << 20, seed=1)
        e9 = Engine(mode="tree", precision="f32", theta=0.5, device=local, per_wave=True)
        e9.upload(pw.pos, pw.vel, pw.mass)
        e9.set_stream(torch.cuda.current_stream().cuda_stream)
        s9 = timed_steps(lambda: e9.step(1), sync, barrier, 2, 20)
        st9 = e9.step_timed()
        extra["bh_theta0.5_fp32_plummer_1M_per_wave_criterion"] = {
            "ms_per_step": 1e3 * s9 / 20, "ms_force": st9.ms_force, "interactions_per_step": int(st9.interactions),
            "interactions_per_s": st9.interactions * 20 / s9,
            "note": "NBE_WALK_PER_WAVE: a cell is opened for all 128 bodies of a wavefront when one of them would open it"}
        e9.close()
        # ... and at theta = 0.6, where its error against the fp64 direct sum equals the per-body walk's at theta = 0.5
        # (median 6.2e-4 against 5.8e-4, p99 2.6e-3 against 3.1e-3: profiles/r01_theta_study.txt)
        e9 = Engine(mode="tree", precision="f32", theta=0.6, device=local, per_wave=True)
        e9.upload(pw.pos, pw.vel, pw.mass)
        e9.set_stream(torch.cuda.current_stream().cuda_stream)
        s9 = timed_steps(lambda: e9.step(1), sync, barrier, 2, 20)
        extra["bh_theta0.6_fp32_plummer_1M_per_wave_criterion"] = {
            "ms_per_step": 1e3 * s9 / 20, "steps_per_s": 20 / s9,
            "note": "same accuracy as the per-body walk at theta 0.5"}
        e9.close()
        # the headline cloud has equal masses, for which the pair loop leaves the mass out; the same cloud with masses
        # varied by 0.1 % takes the general pair loop (one more packed multiply per pair of interactions)
        pg = physics.plummer(1 << 20, seed=1)
        e8 = Engine(mode="direct", precision="f32", device=local)
        e8.upload(pg.pos, pg.vel, pg.mass * (1.0 + 1e-3 * np.sin(np.arange(1 << 20))))
        kms8 = e8.time_force_kernel(3)
        extra["direct_fp32_plummer_1M_unequal_masses"] = {
            "force_kernel_ms": kms8, "interactions_per_s": (1 << 20) * ((1 << 20) - 1) / (kms8 * 1e-3),
            "force_kernel_frac_of_fp32_valu_peak": (1 << 20) * ((1 << 20) - 1) * FLOP_PER_INTERACTION / (kms8 * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS}
        e8.close()
        # benchmark configuration 1: the reference's default scene (main.go:32-37: 1000 bodies + 2 cores), tree, theta = 1
        g = physics.makebodies(1000, seed=1)
        e5 = Engine(mode="tree", precision="f32", theta=1.0, device=local)
        e5.upload(g.pos, g.vel, g.mass)
        e5.set_stream(torch.cuda.current_stream().cuda_stream)
        s5 = timed_steps(lambda: e5.step(1), sync, barrier, 10, 100)
        extra["default_scene_1002_tree_theta1"] = {"gpu_ms_per_step": 1e3 * s5 / 100}
        e5.close()
        # the same scene with the Go program's own arithmetic (running cell centres, last n mod 4 bodies unwalked):
        # what the CPU port below computes, number for number
        e5 = Engine(mode="tree", precision="f64", theta=1.0, device=local, ref_com=True, drop_tail=True)
        e5.upload(g.pos, g.vel, g.mass)
        e5.set_stream(torch.cuda.current_stream().cuda_stream)
        s5 = timed_steps(lambda: e5.step(1), sync, barrier, 10, 100)
        extra["default_scene_1002_tree_theta1"]["gpu_ms_per_step_reference_quirks_fp64"] = 1e3 * s5 / 100
        e5.close()
        if rank == 0 and not args.no_cpu_baseline:
            from oracle import oracle as O
            t0 = time.perf_counter()
            O.step(g.mass, g.pos, g.vel, nsteps=100, mode="tree", com_mode="ref", theta=1.0, drop_tail=True, groups=4)
            extra["default_scene_1002_tree_theta1"]["cpu_port_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / 100
            extra["default_scene_1002_tree_theta1"]["cpu_port_note"] = "C port of main.go:129-178 as written: serial build, 4 walk threads, reference COM update, last N mod 4 bodies unwalked"

    if not args.no_extra and world == 1:
        # benchmark configuration 5 at its full size on one GPU: two 8,388,608-body galaxies on the reference's approach
        # course (main.go:35-36), fp32 pair terms with fp64 sums and an fp64 integrator, softened (eps = 1 m), 1000
        # fused steps, with the drift it is judged by at 100 and 1000 steps
        gp = physics.galaxy_pair(1 << 23, seed=1)
        e6 = Engine(mode="tree", precision="mixed", theta=0.5, eps=1.0, device=local)
        e6.upload(gp.pos, gp.vel, gp.mass)
        e6.set_stream(torch.cuda.current_stream().cuda_stream)
        ke0, pe0, p0 = e6.energy()
        pscale = float((gp.mass * np.linalg.norm(gp.vel, axis=1)).sum())
        del gp
        drift = {}
        s6, done = 0.0, 0
        for upto in (100, 1000):
            s6 += timed_steps(lambda: e6.step(1), sync, barrier, 0, upto - done)
            done = upto
            ke1, pe1, p1 = e6.energy()
            drift[f"after_{upto}_steps"] = {"energy_drift_rel": abs((ke1 + pe1) - (ke0 + pe0)) / abs(ke0 + pe0),
                                            "momentum_drift_rel": float(np.linalg.norm(p1 - p0)) / pscale}
        e6.step(1)                     # the energy left a built tree behind; time a step that sorts and builds itself
        st6 = e6.step_timed()
        extra["galaxy_pair_2x8M_bh_mixed_eps1"] = {
            "bodies": 1 << 24, "ms_per_step": 1e3 * s6 / done, "steps_per_s": done / s6, "steps": done, "drift": drift,
            "bodies_left": int(st6.n_alive), "ms_sort": st6.ms_sort, "ms_build": st6.ms_build, "ms_force": st6.ms_force,
            "stated_bound": "|dE/E0| <= 2e-3 after 100 steps (asserted in tests/test_gpu_config5.py), <= 1e-2 after 1000; "
                            "|dP|/sum m|v| <= 1e-5 after 100, <= 1e-3 after 1000",
            "note": "potential summed over the nodes the walk accepts (theta 0.5) at every checkpoint"}
        e6.close()
        # the same scene on the key-range-distributed tree, eight ranks side by side on this GPU
        gp = physics.galaxy_pair(1 << 23, seed=1)
        extra["galaxy_pair_2x8M_bh_mixed_eps1_distributed_tree_8_ranks_on_one_gpu"] = eight_ranks_on_one_gpu(
            gp, extra["galaxy_pair_2x8M_bh_mixed_eps1"]["ms_per_step"], precision="mixed", theta=0.5, eps=1.0)
        del gp
        # the reference's own precision (fp64 throughout) and the mixed mode on the headline cloud: direct-sum force
        # kernel and Barnes-Hut step, each with its fraction of the vector peak of its arithmetic
        pd = physics.plummer(1 << 20, seed=1)
        npairs = (1 << 20) * ((1 << 20) - 1)
        for prec, key, peak in (("f64", "direct_f64_plummer_1M", 78.6), ("mixed", "direct_mixed_plummer_1M", FP32_VALU_PEAK_TFLOPS)):
            ed = Engine(mode="direct", precision=prec, device=local)
            ed.upload(pd.pos, pd.vel, pd.mass)
            kd = ed.time_force_kernel(2)
            extra[key] = {"force_kernel_ms": kd, "interactions_per_s": npairs / (kd * 1e-3),
                          "vector_peak_tflops": peak,
                          "force_kernel_frac_of_vector_peak": npairs * FLOP_PER_INTERACTION / (kd * 1e-3) / 1e12 / peak}
            ed.close()
        for prec, key, peak in (("f64", "bh_theta0.5_f64_plummer_1M", 78.6), ("mixed", "bh_theta0.5_mixed_plummer_1M", FP32_VALU_PEAK_TFLOPS)):
            et = Engine(mode="tree", precision=prec, theta=0.5, device=local)
            et.upload(pd.pos, pd.vel, pd.mass)
            et.set_stream(torch.cuda.current_stream().cuda_stream)
            st_ = timed_steps(lambda: et.step(1), sync, barrier, 2, 10)
            stt = et.step_timed()
            kt = et.time_force_kernel(3)
            it_ = et.stats().interactions
            extra[key] = {"ms_per_step": 1e3 * st_ / 10, "steps_per_s": 10 / st_, "ms_sort": stt.ms_sort, "ms_build": stt.ms_build,
                          "ms_force": stt.ms_force, "interactions_per_step": int(it_), "force_kernel_ms": kt,
                          "vector_peak_tflops": peak,
                          "force_kernel_frac_of_vector_peak": it_ * FLOP_PER_INTERACTION / (kt * 1e-3) / 1e12 / peak}
            et.close()
        del pd
        # the 1M-body step with collisions on (radius 2 m, physics.go:20): search, queue, merge and update after the walk
        pl = physics.plummer(1 << 20, seed=1)
        e7 = Engine(mode="tree", precision="f32", theta=0.5, collisions=True, device=local)
        e7.upload(pl.pos, pl.vel, pl.mass)
        e7.set_stream(torch.cuda.current_stream().cuda_stream)
        s7 = timed_steps(lambda: e7.step(1), sync, barrier, 2, 20)
        st7 = e7.stats()
        extra["bh_theta0.5_fp32_plummer_1M_collisions_on"] = {"ms_per_step": 1e3 * s7 / 20, "merges_in_22_steps": int(st7.merges),
                                                             "bodies_left": int(st7.n_alive)}
        e7.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_direct(args.cpu_sample) if headline_mode == "direct" else None
        try:
            tb = cpu_baseline_tree(1 << 20, 0.5)
            if headline_mode == "tree":
                cpu = {"value": tb["interactions_per_s"], "unit": "interactions/s", "cores": 4, "kind": "port", "sample": tb["sample"]}
            extra["cpu_tree_path_1M"] = tb
            # the same pass with one walk thread per core of this job's CPU share (the reference is fixed at four)
            cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            from nbody_amd import physics as _ph
            from oracle import oracle as O
            b = _ph.plummer(1 << 20, seed=1)
            t0 = time.perf_counter()
            _, _, st = O.forces(b.mass, b.pos, mode="tree", com_mode="true", theta=0.5, drop_tail=False, groups=cores)
            dt_all = time.perf_counter() - t0
            extra["cpu_tree_path_1M_all_cores"] = {"steps_per_s": 1.0 / dt_all, "interactions_per_s": st.interactions / dt_all,
                                                   "cores": cores, "kind": "port", "build_s": st.build_seconds, "walk_s": st.walk_seconds,
                                                   "sample": f"as cpu_tree_path_1M with {cores} walk threads instead of the reference's 4"}
        except Exception as ex:  # the baseline must never take the benchmark down
            log("cpu tree baseline failed:", ex)


    def distributed_tree_extra():
        """The two 16M-body scenes on the key-range-distributed tree, one rank per GPU: the fp32 Plummer sphere
        (configuration 4) and the two galaxies in mixed precision (configuration 5).  It is the newest multi-process path
        and runs last: a watchdog prints the line without it should its collectives not come back."""
        from nbody_amd.disttree import DistRank, DistributedTree, TorchComm
        m = 1 << 24
        per = (m + world - 1) // world
        for key, b, kw in (("bh_theta0.5_fp32_plummer_16M_distributed_tree", physics.plummer(m, seed=1), dict(precision="f32", theta=0.5)),
                           ("galaxy_pair_2x8M_bh_mixed_eps1_distributed_tree", physics.galaxy_pair(1 << 23, seed=1),
                            dict(precision="mixed", theta=0.5, eps=1.0))):
            r = rank
            dr = DistRank(r, world, device=local, **kw)
            dr.upload(int(per * 1.25) + 4096, b.pos[r * per:(r + 1) * per], b.vel[r * per:(r + 1) * per], b.mass[r * per:(r + 1) * per], r * per)
            dr.use_torch_stream()
            del b
            ranks = [dr]
            tree = DistributedTree(ranks, TorchComm())
            tree.distribute()
            sec = timed_steps(lambda: tree.step(1), sync, barrier, 2, 5)      # (the first step moves every body to its owner)
            runs = [tree.step_profiled() for _ in range(5)]
            # (a fixed list of phases, alike on every rank: the reduction below goes by position; "rebalance" only shows in
            # a run in which new splitters were chosen)
            keys = ["classify", "migrate", "sort_build", "export", "exchange_assemble", "walk", "rebalance"]
            acc = {k: float(np.median([t.get(k, 0.0) for t in runs])) for k in keys}
            live = float(ranks[0].engine.stats().n_alive)
            if world > 1:
                t = torch.tensor([sec] + [acc[k] for k in sorted(acc)], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                sec = float(t[0].item())
                acc = {k: float(v) for k, v in zip(sorted(acc), t[1:].tolist())}
                t = torch.tensor([live, -live], device=red_dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                live = [float(t[0].item()), -float(t[1].item())]
            ref = one_gpu_reference("bh_theta0.5_fp32_plummer_16M" if "plummer" in key else "galaxy_pair_2x8M_bh_mixed_eps1")
            extra[key] = {
                "steps_per_s": 5 / sec, "ms_per_step": 1e3 * sec / 5, "phase_ms_max_over_ranks": acc,
                "steps_per_s_over_one_gpu": {"ratio": (5 / sec) / ref["value"], "one_gpu": ref} if ref else None,
                "bodies_per_rank_max_min": live, "imported_nodes_this_rank": tree.last_import_nodes[0],
                "migrated_bodies_this_rank": tree.last_migrated[0]}
            for rr in ranks:
                rr.close()

    def emit():
        if rank == 0:
            os.write(json_fd, (json.dumps(line()) + "\n").encode())

    if (world > 1 or force_dist) and not args.no_extra:
        import threading
        finished = threading.Event()

        def watchdog():
            if not finished.wait(float(os.environ.get("NBODY_BENCH_DIST_TIMEOUT", "300"))):
                extra.setdefault("bh_theta0.5_fp32_plummer_16M_distributed_tree", {"error": "did not finish in time; the rest of the line is complete"})
                extra.setdefault("galaxy_pair_2x8M_bh_mixed_eps1_distributed_tree", {"error": "did not finish in time; the rest of the line is complete"})
                emit()
                os._exit(3)                   # the line is out; the exit code says that a collective did not come back
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            distributed_tree_extra()
        except Exception as ex:
            extra.setdefault("bh_theta0.5_fp32_plummer_16M_distributed_tree", {"error": repr(ex)})
            extra.setdefault("galaxy_pair_2x8M_bh_mixed_eps1_distributed_tree", {"error": repr(ex)})
            log("distributed tree extra failed:", ex)
        finished.set()

    def line():
        return {"metric": "body-body interactions/sec", "value": value, "unit": "interactions/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * sec / args.steps, "steps_per_s": args.steps / sec,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": workload, "bodies": n, "ic": "Plummer sphere a=2e3 m, M=1e9 kg, seed 1",
                          "dt": 3600.0, "eps": 0.0, "parallelism": f"targets sharded over {world} GPU(s), all-gather of pos+vel per step" if world > 1 else "single GPU"},
               "roofline": roofline, "cpu_baseline": cpu, "extra": extra}

    emit()
    if world > 1 or force_dist:
        # a rank that failed above while the others were inside a collective would wait here for ever
        import threading
        failed = any(isinstance(v, dict) and "error" in v for v in extra.values())
        threading.Timer(120.0, lambda: os._exit(3)).start()      # (the line is out already; a barrier nobody joins must not pass for success)
        dist.barrier()
        dist.destroy_process_group()
        os._exit(3 if failed else 0)


if __name__ == "__main__":
    main()
