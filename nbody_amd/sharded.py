"""Multi-GPU stepping: one process and one engine per GPU, targets sharded, sources replicated.

The reference's only data-parallel strategy gives each of four goroutines a contiguous quarter of the
bodies against a shared read-only tree (main.go:134-149).  Here a rank is such a worker: it holds all n
bodies as sources, advances only its contiguous range of targets, and the ranks exchange the advanced
ranges with one all-gather per state array per step (torch.distributed: backend "nccl" is RCCL over xGMI;
"gloo" is used for CPU rehearsals and stages device buffers through the host).  In tree mode every rank
sorts and builds the same tree over the same replicated state, so the sorted order, and with it the meaning
of a target range, is identical everywhere.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is part of the image
    torch = None
    dist = None


def shard_range(n, rank, world, align=256):
    """Contiguous range of ceil(n/world) targets per rank, rounded up to a multiple of 256 (the last ranks may get
    fewer or none).  Ranges that start at multiples of 256 give the wavefronts of the tree walk the same bodies as a
    one-GPU run, which the per-wave criterion needs to decide as one GPU does (it opens cells for aligned groups of 128
    bodies); the per-body walk gives every body the same bits wherever the ranges begin, in every precision."""
    per = (n + world - 1) // world
    per = (per + align - 1) // align * align
    first = min(rank * per, n)
    return first, min(per, n - first), per


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


def _wrap(ptr, n, elem_bytes, device):
    t = torch.as_tensor(_DevicePtr(ptr, n * elem_bytes), device=device)
    dtype = torch.float32 if elem_bytes == 16 else torch.float64
    return t.view(dtype).view(n, 4)


class EngineShard:
    """Adapter between an nbody_amd.Engine and the stepper: exposes the engine's next-state buffers as
    torch tensors that alias device memory."""

    def __init__(self, engine, device_index=0, use_current_stream=True):
        self.engine = engine
        self.n = engine.n
        self.device = torch.device("cuda", device_index)
        if use_current_stream:
            engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._views = {}
        # collisions (tree mode): advance leaves accelerations, the ranks exchange those, commit merges and updates
        self.collisions = bool(engine.cfg.collisions)

    def set_shard(self, first, count):
        self.engine.set_shard(first, count)

    def prepare(self):
        self.engine.shard_prepare()

    def sort_ahead(self):
        """The part of the next step's sort that needs positions only (nbe_shard_sort)."""
        self.engine.shard_sort()

    def advance(self):
        self.engine.shard_advance()

    def advance_range(self, first, count):
        self.engine.shard_advance_range(first, count)

    def commit(self):
        self.engine.shard_commit()

    def _view(self, ptr, elem_bytes):
        if ptr not in self._views:
            self._views[ptr] = _wrap(ptr, self.n, elem_bytes, self.device)
        return self._views[ptr]

    def next_state(self):
        ds = self.engine.device_state()
        return [self._view(ptr, ds.elem_bytes) for ptr in (ds.pos_mass_next, ds.vel_next)]

    def accelerations(self):
        ds = self.engine.device_state()
        return self._view(ds.acc, ds.elem_bytes)


class ShardedStepper:
    """Drives one rank's shard.  `shard` is an EngineShard (or anything with the same methods)."""

    def __init__(self, shard, rank=None, world=None, group=None, always_gather=False, align=256, split=False):
        """always_gather: run the collectives even with a single rank (exercises the RCCL path on one GPU).
        align: target ranges start at multiples of this many bodies (see shard_range).
        split: give every rank two ranges, one in each half of the sorted order, and exchange the positions of the
        first while the second is computed (needs n divisible by 2 * world * align and a shard with advance_range);
        None = do so whenever possible on the RCCL backend.  Off by default: a walk launch ends with a tail of about
        one wavefront's lifetime (~1 ms at eight waves per SIMD) in which the GPU drains, so two launches of 1M bodies
        cost 4.3-4.4 ms against 4.15 ms for one of 2M (measured on one MI355X, 16M-body cloud), which gives back half
        of the ~0.45 ms of exchange that the split hides."""
        self.align = align
        self.always_gather = always_gather
        self.shard = shard
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n = shard.n
        self.first, self.count, self.per = shard_range(self.n, self.rank, self.world, align)
        shard.set_shard(self.first, self.count)
        self.even = self.per * self.world == self.n
        self.backend = dist.get_backend(group) if (self.world > 1 or always_gather) else None
        self._pad = {}
        self.collisions = bool(getattr(shard, "collisions", False))
        if self.collisions and split:
            raise ValueError("split exchanges next-state rows; with collisions the ranks exchange accelerations")
        can_split = (hasattr(shard, "advance_range") and not self.collisions and self.n > 0 and self.n % (2 * self.world * align) == 0
                     and (self.world > 1 or always_gather))
        if split is None:
            split = can_split and self.backend == "nccl"
        if split and not can_split:
            raise ValueError("split needs n divisible by 2 * world * align and a shard with advance_range")
        self.split = bool(split)
        if self.split:
            # rank r owns [r*q, (r+1)*q) of the first half of the sorted order and the same slice of the second half:
            # each half is exchanged by one in-place all-gather, the first one while the second range is computed
            self.half = self.n // 2
            self.q = self.half // self.world
            self.ranges = [(k * self.half + self.rank * self.q, self.q) for k in range(2)]
            self.first, self.count = self.ranges[0]
        else:
            self.ranges = [(self.first, self.count)]

    def _gather(self, full):
        """All-gather the ranks' row ranges of `full` (n x 4) in place."""
        if self.world == 1 and not self.always_gather:
            return
        host_stage = self.backend == "gloo" and full.is_cuda
        work = full.cpu() if host_stage else full
        if self.even:
            mine = work[self.first:self.first + self.count]
            dist.all_gather_into_tensor(work, mine, group=self.group)
        else:
            key = (work.dtype, work.device)
            if key not in self._pad:
                self._pad[key] = (torch.zeros(self.per * self.world, 4, dtype=work.dtype, device=work.device),
                                  torch.zeros(self.per, 4, dtype=work.dtype, device=work.device))
            padded, mine = self._pad[key]
            mine[:self.count] = work[self.first:self.first + self.count]
            dist.all_gather_into_tensor(padded, mine, group=self.group)
            for r in range(self.world):
                f, c, _ = shard_range(self.n, r, self.world, self.align)
                if c:
                    work[f:f + c] = padded[r * self.per:r * self.per + c]
        if host_stage:
            full.copy_(work)

    def _gather_async(self, full):
        """The same exchange as a non-blocking collective, where the backend runs it beside the compute stream
        (RCCL, device buffers, equal ranges); returns the work handle, or None after a blocking exchange."""
        if self.world == 1 and not self.always_gather:
            return None
        if self.backend != "nccl" or not self.even or not full.is_cuda:
            self._gather(full)
            return None
        return dist.all_gather_into_tensor(full, full[self.first:self.first + self.count], group=self.group, async_op=True)

    def _gather_half(self, full, k, async_op):
        """All-gather half k of `full` (the ranks' slices of it are equal and in rank order) in place; returns a work
        handle when the exchange runs beside the compute stream, else None."""
        view = full[k * self.half:(k + 1) * self.half]
        if self.backend == "gloo" and view.is_cuda:
            work = view.cpu()
            dist.all_gather_into_tensor(work, work[self.rank * self.q:(self.rank + 1) * self.q].clone(), group=self.group)
            view.copy_(work)
            return None
        mine = view[self.rank * self.q:(self.rank + 1) * self.q]
        if async_op and self.backend == "nccl":
            return dist.all_gather_into_tensor(view, mine, group=self.group, async_op=True)
        dist.all_gather_into_tensor(view, mine if self.backend == "nccl" else mine.clone(), group=self.group)
        return None

    def _step_split(self):
        """One step with two ranges per rank: the positions of the first range travel while the second is computed."""
        self.shard.prepare()
        pos, vel = self.shard.next_state()
        pending = []
        for k, (first, count) in enumerate(self.ranges):
            self.shard.advance_range(first, count)
            pending.append(self._gather_half(pos, k, async_op=True))
        for w in pending:
            if w is not None:
                w.wait()                # stream dependencies, not host waits
        vpending = [self._gather_half(vel, k, async_op=hasattr(self.shard, "sort_ahead")) for k in range(2)]
        self.shard.commit()
        if any(w is not None for w in vpending):
            self.shard.sort_ahead()
            for w in vpending:
                if w is not None:
                    w.wait()

    def step(self, nsteps=1):
        """Per step: build (replicated), advance the own range, exchange positions, then velocities.  Only the
        positions are needed to start the next step (keys and radix passes), so the velocity exchange runs beside
        that sort when the shard offers `sort_ahead`; the stream waits for it before the state is reordered.
        With `split` the position exchange itself is half hidden behind the force pass (_step_split)."""
        if self.split:
            for _ in range(nsteps):
                self._step_split()
            return
        if self.collisions:
            # the ranks share out the force pass; every rank then merges and updates all bodies alike (the search needs
            # the tree of this step and the merges change neighbours, so nothing of the next step can run ahead)
            for _ in range(nsteps):
                self.shard.prepare()
                self.shard.advance()
                self._gather(self.shard.accelerations())
                self.shard.commit()
            return
        for _ in range(nsteps):
            self.shard.prepare()
            self.shard.advance()
            pos, vel = self.shard.next_state()
            self._gather(pos)
            pending = self._gather_async(vel) if hasattr(self.shard, "sort_ahead") else (self._gather(vel), None)[1]
            self.shard.commit()
            if pending is not None:
                self.shard.sort_ahead()
                pending.wait()          # a stream dependency, not a host wait

    def step_profiled(self):
        """One step with every phase bracketed by events on the compute stream and the exchanges run blocking, one
        after the other (no overlap): returns the milliseconds of build (sort + tree, replicated), advance (forces +
        update of the own range), the position exchange and the velocity exchange.  For diagnosis only."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        self.shard.prepare()
        ev[1].record()
        if self.split:
            for first, count in self.ranges:
                self.shard.advance_range(first, count)
        else:
            self.shard.advance()
        ev[2].record()
        if self.collisions:
            self._gather(self.shard.accelerations())
            ev[3].record()
            ev[4].record()
            self.shard.commit()
            torch.cuda.synchronize()
            return {"build": ev[0].elapsed_time(ev[1]), "advance": ev[1].elapsed_time(ev[2]),
                    "gather_accelerations": ev[2].elapsed_time(ev[3])}
        pos, vel = self.shard.next_state()
        if self.split:
            for k in range(2):
                self._gather_half(pos, k, async_op=False)
            ev[3].record()
            for k in range(2):
                self._gather_half(vel, k, async_op=False)
        else:
            self._gather(pos)
            ev[3].record()
            self._gather(vel)
        ev[4].record()
        self.shard.commit()
        torch.cuda.synchronize()
        names = ("build", "advance", "gather_positions", "gather_velocities")
        return {n: ev[k].elapsed_time(ev[k + 1]) for k, n in enumerate(names)}
