"""Key-range-distributed Barnes-Hut stepping: every body lives on one GPU, every GPU builds the tree of its own
bodies, and the GPUs exchange only the bodies that change owner and the pruned subtrees the others need.

The reference walks one octree of all bodies with a fan-out over workers (main.go:130-149); this module keeps that
semantics (every body visits the nodes it would visit in the one octree, in the same order) while no GPU ever holds
the whole octree.  The device work is in the engine (nbe_dist_* in include/nbody_engine.h, nbody_amd/csrc/dist.hip);
this module moves the bytes: it is written once, for a list of ranks that live in this process, and a communicator
either copies between them (LocalComm: several ranks on one GPU, for tests) or runs the collectives of
torch.distributed (TorchComm: one rank per process, RCCL over xGMI; gloo stages through the host).
"""
import ctypes as C

import numpy as np

from .engine import Engine, EngineError, load_library

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover - torch is part of the image
    torch = None
    dist = None

MAX_FRAGMENTS = 2048
MAX_BOXES = 8
FRAGMENT = np.dtype([("key", "<u8"), ("level", "<i4"), ("start", "<i4"), ("nodes", "<i4"), ("reserved", "<i4"),
                     ("sum", "<u8", (8,))])
assert FRAGMENT.itemsize == 88


class DistBuffers(C.Structure):
    _fields_ = [("outbox", C.c_void_p), ("inbox", C.c_void_p), ("export_nodes", C.c_void_p), ("import_nodes", C.c_void_p),
                ("slots", C.c_uint64), ("export_cap", C.c_uint64), ("import_cap", C.c_uint64),
                ("record_bytes", C.c_uint32), ("node_bytes", C.c_uint32)]


_declared = False


def _declare(lib):
    global _declared
    if _declared:
        return
    vp, dp, u64p = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)
    lib.nbe_dist_upload.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t, dp, dp, dp, C.c_uint64]
    lib.nbe_dist_sample_keys.argtypes = [vp, C.c_size_t, u64p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nbe_dist_set_splitters.argtypes = [vp, u64p]
    lib.nbe_dist_set_total_mass.argtypes = [vp, C.c_double]
    lib.nbe_dist_set_radius.argtypes = [vp, dp, C.c_double]
    lib.nbe_dist_buffers.argtypes = [vp, C.POINTER(DistBuffers)]
    lib.nbe_dist_classify.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.nbe_dist_receive.argtypes = [vp, C.c_size_t]
    lib.nbe_dist_room.argtypes = [vp, u64p]
    lib.nbe_dist_unclassify.argtypes = [vp]
    lib.nbe_dist_build.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int), dp, C.POINTER(C.c_int), u64p]
    lib.nbe_dist_export.argtypes = [vp, vp, C.POINTER(C.c_int), dp, C.POINTER(C.c_int), C.POINTER(C.c_int32), u64p]
    lib.nbe_dist_assemble.argtypes = [vp, C.POINTER(C.c_int32), u64p]
    lib.nbe_dist_advance.argtypes = [vp]
    lib.nbe_dist_energy.argtypes = [vp, dp, dp, dp]
    lib.nbe_dist_layout.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int64)]
    lib.nbe_dist_download.argtypes = [vp, u64p, dp, dp, dp, C.POINTER(C.c_size_t)]
    lib.nbe_dist_get_radius.argtypes = [vp, u64p, dp, C.POINTER(C.c_size_t)]
    lib.nbe_dist_accel.argtypes = [vp, u64p, dp, C.POINTER(C.c_size_t)]
    for name in DIST_EXPORTS:
        getattr(lib, name).restype = C.c_int
    _declared = True


TOP_CELL = np.dtype([("dst", "<i4"), ("skip", "<i4"), ("level", "<i4"), ("reserved", "<i4"), ("sum", "<u8", (8,))])


def layout(all_frags, sizes):
    """nbe_dist_layout: (spanning cells, first index of every fragment, total nodes) of the global pre-order made of the
    given fragments (key order) with the given node counts.  Host only."""
    lib = load_library()
    _declare(lib)
    all_frags = np.ascontiguousarray(all_frags, dtype=FRAGMENT)
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    n = len(all_frags)
    tops = np.zeros(max(n, 1), dtype=TOP_CELL)
    dst = np.zeros(max(n, 1), dtype=np.int32)
    nt, total = C.c_int(0), C.c_int64(0)
    rc = lib.nbe_dist_layout(all_frags.ctypes.data, n, sizes.ctypes.data_as(C.POINTER(C.c_int32)), tops.ctypes.data, len(tops),
                             C.byref(nt), dst.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(total))
    if rc != 0:
        raise EngineError(f"nbody_engine error {rc}: {lib.nbe_last_error().decode()}")
    return tops[:nt.value].copy(), dst[:n].copy(), total.value


DIST_EXPORTS = ["nbe_dist_layout", "nbe_dist_energy", "nbe_dist_set_total_mass", "nbe_dist_set_radius", "nbe_dist_upload", "nbe_dist_sample_keys", "nbe_dist_set_splitters", "nbe_dist_buffers", "nbe_dist_classify",
                "nbe_dist_receive", "nbe_dist_build", "nbe_dist_export", "nbe_dist_assemble", "nbe_dist_advance",
                "nbe_dist_download", "nbe_dist_get_radius", "nbe_dist_room", "nbe_dist_unclassify", "nbe_dist_accel"]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class _DevicePtr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


class DistRank:
    """One rank of the distributed tree: an Engine (tree mode) plus the nbe_dist_* calls."""

    def __init__(self, rank, world, device=0, **engine_kw):
        self.rank, self.world = int(rank), int(world)
        self.engine = Engine(mode="tree", device=device, **engine_kw)
        self._lib = load_library()
        _declare(self._lib)
        self._h = self.engine._h
        self.device_index = device
        self.capacity = 0
        self._views = {}

    def close(self):
        self.engine.close()

    def _check(self, rc):
        self.engine._check(rc)

    def upload(self, capacity, pos, vel, mass, first_index):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        mass = np.ascontiguousarray(mass, dtype=np.float64).reshape(-1)
        self._check(self._lib.nbe_dist_upload(self._h, self.rank, self.world, int(capacity), mass.shape[0], _dp(pos), _dp(vel),
                                              _dp(mass), int(first_index)))
        self.capacity = int(capacity)
        self.local_mass = float(mass.sum())
        self.engine.n = self.capacity
        b = DistBuffers()
        self._check(self._lib.nbe_dist_buffers(self._h, C.byref(b)))
        self.buffers = b
        self._views = {}

    def use_torch_stream(self):
        """Queue the engine's work on torch's current stream, so that it is ordered with torch's copies and collectives."""
        self.engine.set_stream(torch.cuda.current_stream(torch.device("cuda", self.device_index)).cuda_stream)

    def bytes_view(self, which):
        """A torch uint8 tensor over one of the engine's exchange buffers (no copy)."""
        if which not in self._views:
            b = self.buffers
            ptr, n = {"outbox": (b.outbox, b.slots * b.record_bytes), "inbox": (b.inbox, b.slots * b.record_bytes),
                      "export": (b.export_nodes, b.export_cap * b.node_bytes),
                      "import": (b.import_nodes, b.import_cap * b.node_bytes)}[which]
            self._views[which] = torch.as_tensor(_DevicePtr(ptr, int(n)), device=torch.device("cuda", self.device_index))
        return self._views[which]

    def sample_keys(self, max_samples=4096):
        keys = np.zeros(max_samples, dtype=np.uint64)
        cnt, live = C.c_size_t(0), C.c_size_t(0)
        self._check(self._lib.nbe_dist_sample_keys(self._h, max_samples, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   C.byref(cnt), C.byref(live)))
        return keys[:cnt.value], live.value

    def set_radius(self, radius, radius_cap=32.0):
        """Collisions on: the radii of the uploaded bodies (None keeps the reference's 2 m) and the largest radius any body
        of the run may reach, the same on every rank; right after upload()."""
        r = None if radius is None else np.ascontiguousarray(radius, dtype=np.float64).reshape(-1)
        self._check(self._lib.nbe_dist_set_radius(self._h, None if r is None else _dp(r), float(radius_cap)))

    def set_total_mass(self, total):
        """The run's total mass (all ranks, the same value everywhere): the quantum of the exact cell sums follows from it."""
        self._check(self._lib.nbe_dist_set_total_mass(self._h, float(total)))

    def set_splitters(self, split):
        split = np.ascontiguousarray(split, dtype=np.uint64)
        assert split.shape[0] == self.world + 1
        self._check(self._lib.nbe_dist_set_splitters(self._h, split.ctypes.data_as(C.POINTER(C.c_uint64))))

    def classify(self):
        counts = np.zeros(self.world, dtype=np.uint32)
        self._check(self._lib.nbe_dist_classify(self._h, counts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return counts

    def room(self):
        """After classify(): how many records this rank can take in the receive() that follows."""
        v = C.c_uint64(0)
        self._check(self._lib.nbe_dist_room(self._h, C.byref(v)))
        return int(v.value)

    def unclassify(self):
        """Undoes the last classify() (before any receive()): the leavers return to their slots."""
        self._check(self._lib.nbe_dist_unclassify(self._h))

    def receive(self, k):
        self._check(self._lib.nbe_dist_receive(self._h, int(k)))

    def build(self):
        frags = np.zeros(MAX_FRAGMENTS, dtype=FRAGMENT)
        cnt, nb = C.c_int(0), C.c_int(0)
        boxes = np.zeros((MAX_BOXES, 6))
        live = C.c_uint64(0)
        self._check(self._lib.nbe_dist_build(self._h, frags.ctypes.data, MAX_FRAGMENTS, C.byref(cnt), _dp(boxes), C.byref(nb),
                                             C.byref(live)))
        return frags[:cnt.value].copy(), boxes[:nb.value].copy(), live.value

    def export(self, all_frags, frag_counts, boxes):
        """boxes: per rank, its [k, 6] boxes."""
        all_frags = np.ascontiguousarray(all_frags, dtype=FRAGMENT)
        frag_counts = np.ascontiguousarray(frag_counts, dtype=np.int32)
        padded = np.zeros((self.world, MAX_BOXES, 6))
        box_counts = np.zeros(self.world, dtype=np.int32)
        for r, bx in enumerate(boxes):
            bx = np.asarray(bx, dtype=np.float64).reshape(-1, 6)
            padded[r, :len(bx)] = bx
            box_counts[r] = len(bx)
        nf = int(frag_counts[self.rank])
        sizes = np.zeros((self.world, max(nf, 1)), dtype=np.int32)
        offsets = np.zeros(self.world + 1, dtype=np.uint64)
        self._check(self._lib.nbe_dist_export(self._h, all_frags.ctypes.data, frag_counts.ctypes.data_as(C.POINTER(C.c_int)),
                                              _dp(padded), box_counts.ctypes.data_as(C.POINTER(C.c_int)),
                                              sizes.ctypes.data_as(C.POINTER(C.c_int32)),
                                              offsets.ctypes.data_as(C.POINTER(C.c_uint64))))
        return sizes.reshape(-1)[:self.world * nf].reshape(self.world, nf), offsets

    def assemble(self, import_sizes, import_offsets):
        import_sizes = np.ascontiguousarray(import_sizes, dtype=np.int32)
        import_offsets = np.ascontiguousarray(import_offsets, dtype=np.uint64)
        self._check(self._lib.nbe_dist_assemble(self._h, import_sizes.ctypes.data_as(C.POINTER(C.c_int32)),
                                                import_offsets.ctypes.data_as(C.POINTER(C.c_uint64))))

    def advance(self):
        self._check(self._lib.nbe_dist_advance(self._h))

    def energy(self):
        """(kinetic, potential share, momentum[3]) of this rank's bodies; needs the assembled tree."""
        ke, pe = C.c_double(), C.c_double()
        mom = np.zeros(3)
        self._check(self._lib.nbe_dist_energy(self._h, C.byref(ke), C.byref(pe), _dp(mom)))
        return ke.value, pe.value, mom

    def download(self):
        n = self.capacity
        idx = np.zeros(n, dtype=np.uint64); pos = np.zeros((n, 3)); vel = np.zeros((n, 3)); mass = np.zeros(n)
        k = C.c_size_t(0)
        self._check(self._lib.nbe_dist_download(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(pos), _dp(vel), _dp(mass),
                                                C.byref(k)))
        k = k.value
        return idx[:k], pos[:k], vel[:k], mass[:k]

    def accel(self):
        """(run-wide indices, accelerations[k, 3]) of this rank's bodies; needs the assembled tree."""
        n = self.capacity
        idx = np.zeros(n, dtype=np.uint64); acc = np.zeros((n, 3))
        k = C.c_size_t(0)
        self._check(self._lib.nbe_dist_accel(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(acc), C.byref(k)))
        return idx[:k.value], acc[:k.value]

    def get_radius(self):
        """Collisions on: (run-wide indices, radii) of this rank's live bodies, in the order of download()."""
        n = self.capacity
        idx = np.zeros(n, dtype=np.uint64); rad = np.zeros(n)
        k = C.c_size_t(0)
        self._check(self._lib.nbe_dist_get_radius(self._h, idx.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(rad), C.byref(k)))
        return idx[:k.value], rad[:k.value]


# ---- communicators --------------------------------------------------------------------------------------------------
class LocalComm:
    """All ranks live in this process (on one GPU or several): exchanges are device copies."""

    def __init__(self, world):
        self.world = world

    def allgather(self, per_rank):
        """per_rank[r] = numpy array of rank r; every rank receives the list of all of them."""
        return [list(per_rank) for _ in range(self.world)]

    def allgather_many(self, per_rank):
        """per_rank[r] = list of numpy arrays of rank r; every rank receives the list (over ranks) of those lists."""
        return [[list(a) for a in per_rank] for _ in range(self.world)]

    def alltoall(self, ranks, which_send, which_recv, send_counts, unit, recv_counts=None):
        """send_counts[r][q] units of `unit` bytes go from rank r's `which_send` buffer (destination after destination)
        to rank q's `which_recv` buffer (source after source).  Returns recv_counts[q][r] (recv_counts, when the caller
        knows them already, saves a communicator with separate processes one exchange)."""
        W = self.world
        recv = [[int(send_counts[r][q]) for r in range(W)] for q in range(W)]
        for q in range(W):
            dst = ranks[q].bytes_view(which_recv)
            at = 0
            for r in range(W):
                k = recv[q][r]
                if k:
                    off = int(sum(int(c) for c in send_counts[r][:q]))
                    src = ranks[r].bytes_view(which_send)
                    if (at + k) * unit > dst.numel():
                        raise EngineError(f"rank {q}: the {which_recv} buffer is too small for {at + k} records")
                    dst[at * unit:(at + k) * unit].copy_(src[off * unit:(off + k) * unit])
                at += k
        return recv


class TorchComm:
    """One rank per process under torch.distributed ("nccl" = RCCL over xGMI; "gloo" stages device buffers through
    the host)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")

    def allgather(self, per_rank):
        return [[got[0] for got in self.allgather_many([[per_rank[0]]])[0]]]

    def _buffers(self, cap):
        """Staging for the small gathers, kept from step to step: a pinned host buffer and a device buffer on either side
        of the collective (pageable memory would be pinned and unpinned by the runtime in every copy, and fresh tensors
        go through the allocator)."""
        if getattr(self, "_cap", 0) != cap:
            on_gpu = self.device.type == "cuda"
            self._hsend = torch.zeros(cap, dtype=torch.uint8, pin_memory=on_gpu)
            self._hrecv = torch.zeros(self.world * cap, dtype=torch.uint8, pin_memory=on_gpu)
            self._dsend = torch.zeros(cap, dtype=torch.uint8, device=self.device) if on_gpu else self._hsend
            self._drecv = torch.zeros(self.world * cap, dtype=torch.uint8, device=self.device) if on_gpu else self._hrecv
            self._hsend_np, self._hrecv_np = self._hsend.numpy(), self._hrecv.numpy()
            self._cap = cap
        return self._hsend_np, self._hrecv_np

    def allgather_many(self, per_rank):
        """One collective for several small arrays: every rank sends [total bytes | byte length of each array | the
        arrays], padded to a capacity that all ranks hold alike; a rank that sees a longer message from anybody doubles
        the capacity (everybody sees the same lengths, so everybody does) and the exchange is repeated."""
        arrays = [np.ascontiguousarray(a) for a in per_rank[0]]
        blobs = [a.view(np.uint8).reshape(-1) for a in arrays]
        head = np.array([0] + [b.size for b in blobs], dtype=np.int64)
        head[0] = 8 * head.size + int(head[1:].sum())
        msg = np.concatenate([head.view(np.uint8)] + blobs)
        cap = getattr(self, "_cap", 0) or 1 << 16
        while True:
            hsend, hrecv = self._buffers(cap)
            k = min(cap, msg.size)
            hsend[:k] = msg[:k]
            if self.device.type == "cuda":
                self._dsend.copy_(self._hsend, non_blocking=True)
                dist.all_gather_into_tensor(self._drecv, self._dsend, group=self.group)
                self._hrecv.copy_(self._drecv, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            else:
                dist.all_gather_into_tensor(self._hrecv, self._hsend, group=self.group)
            out = hrecv.reshape(self.world, cap)
            need = int(max(out[r, :8].view(np.int64)[0] for r in range(self.world)))
            if need <= cap:
                break
            cap = 2 * need
        res = []
        for r in range(self.world):
            lens = out[r, 8:8 * head.size].view(np.int64)
            at = 8 * head.size
            got = []
            for a, n in zip(arrays, lens):
                raw = out[r, at:at + int(n)].copy().view(a.dtype)
                got.append(raw.reshape((-1,) + a.shape[1:]) if a.ndim > 1 else raw)
                at += int(n)
            res.append(got)
        return [res]

    def alltoall(self, ranks, which_send, which_recv, send_counts, unit, recv_counts=None):
        me = ranks[0]
        sc = [int(c) for c in send_counts[0]]
        if recv_counts is None:
            # everybody's counts through the staged gather above: row r, column of this rank
            rows = self.allgather_many([[np.asarray(sc, dtype=np.int64)]])[0]
            rc = [int(rows[r][0][self.rank]) for r in range(self.world)]
        else:
            rc = [int(c) for c in recv_counts[0]]
        src, dst = me.bytes_view(which_send), me.bytes_view(which_recv)
        ns, nr = sum(sc) * unit, sum(rc) * unit
        if nr > dst.numel():
            raise EngineError(f"rank {self.rank}: the {which_recv} buffer is too small for {sum(rc)} records")
        if self.backend == "nccl":
            dist.all_to_all_single(dst[:nr], src[:ns], [k * unit for k in rc], [k * unit for k in sc], group=self.group)
        else:
            hs = src[:ns].cpu()
            hd = torch.zeros(nr, dtype=torch.uint8)
            dist.all_to_all_single(hd, hs, [k * unit for k in rc], [k * unit for k in sc], group=self.group)
            dst[:nr].copy_(hd)
        return [rc]


def choose_splitters(samples, lives, world):
    """Splitters from every rank's evenly spaced sorted keys: each sample of rank r stands for lives[r] / len(samples[r])
    bodies; the k-th splitter is the key where the running weight passes k/world of all bodies.  Bodies with equal keys
    never end up on two ranks (a splitter is a key; everything below it goes left)."""
    keys = np.concatenate([np.asarray(s, dtype=np.uint64) for s in samples]) if samples else np.zeros(0, np.uint64)
    w = np.concatenate([np.full(len(s), lives[r] / max(len(s), 1)) for r, s in enumerate(samples)]) if samples else np.zeros(0)
    split = np.zeros(world + 1, dtype=np.uint64)
    split[world] = np.uint64(1) << np.uint64(63)
    if len(keys) == 0:
        split[1:world] = split[world]
        return split
    order = np.argsort(keys, kind="stable")
    keys, w = keys[order], w[order]
    cum = np.cumsum(w)
    total = cum[-1]
    for k in range(1, world):
        j = int(np.searchsorted(cum, total * k / world, side="left"))
        split[k] = keys[min(j, len(keys) - 1)]
    for k in range(1, world + 1):
        split[k] = max(split[k], split[k - 1])
    return split


class DistributedTree:
    """Steps the ranks that live in this process (all of them with LocalComm, one with TorchComm)."""

    def __init__(self, ranks, comm, rebalance_at=1.10, freeze_gc=True):
        """rebalance_at: new splitters are chosen before a step when the fullest rank holds this many times the mean
        number of bodies (the bodies' distribution over key space drifts as the system evolves), or when a rank has filled
        nine tenths of its slots; None = never.
        freeze_gc: a step makes a few hundred short-lived lists and arrays, which sets off Python's cyclic collector
        every few steps; a full collection walks every object of the process (35 ms with torch loaded, measured: one step
        in eight took 75 ms instead of 40), so the objects that exist when the run starts are moved out of its sight once
        (gc.freeze), after which a collection only looks at what the steps themselves made."""
        self.ranks, self.comm = list(ranks), comm
        self.world = comm.world
        self.timings = None
        self.rebalance_at = rebalance_at
        self.rebalances = 0
        self._prepared = False
        self._want_rebalance = False
        self._have_total = False
        self._freeze_gc = freeze_gc
        self.lives = None

    def distribute(self, max_samples=4096):
        """Chooses the splitters from samples of every rank's keys; the first step then moves every body to its owner."""
        got = [r.sample_keys(max_samples) for r in self.ranks]
        samples = self.comm.allgather([g[0] for g in got])
        lives = self.comm.allgather([np.array([g[1]], dtype=np.int64) for g in got])
        if not self._have_total:
            # every rank adds the ranks' masses in rank order: the same total, and with it the same quantum, everywhere
            masses = self.comm.allgather([np.array([r.local_mass]) for r in self.ranks])
            for k, r in enumerate(self.ranks):
                r.set_total_mass(float(sum(float(a[0]) for a in masses[k])))
            self._have_total = True
        for k, r in enumerate(self.ranks):
            split = choose_splitters(samples[k], [int(a[0]) for a in lives[k]], self.world)
            r.set_splitters(split)
        self.split = split
        if self._freeze_gc:
            import gc
            gc.collect()
            gc.freeze()
            self._freeze_gc = False
        return split

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self._step()

    def energy(self):
        """(kinetic, potential, momentum[3]) of the whole run for the current state (tree potential, softened by eps):
        the sums of the ranks' shares, the same on every rank."""
        if not self._prepared:
            self._prepare()
        mine = [np.array([ke, pe, *mom]) for ke, pe, mom in (r.energy() for r in self.ranks)]
        tot = np.sum(self.comm.allgather(mine)[0], axis=0)
        return float(tot[0]), float(tot[1]), tot[2:5].copy()

    def step_profiled(self):
        """One step with the device drained after every phase: returns the wall-clock milliseconds of each phase summed
        over the ranks of this process (with LocalComm, divide by their number for what one GPU of a real run does)."""
        import time
        self.timings = {}
        torch.cuda.synchronize()
        self._t0 = time.perf_counter()
        self._step()
        t = self.timings
        self.timings = None
        return t

    def _mark(self, name):
        if self.timings is None:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (now - self._t0) * 1e3
        self._t0 = now

    def _step(self):
        if not self._prepared:
            self._prepare()
        for r in self.ranks:
            r.advance()
        self._prepared = False
        self._mark("walk")

    def _prepare(self):
        """Everything of a step but the force pass: migration, local builds, exchange of pruned fragments, assembly."""
        R, W = self.ranks, self.world
        if self._want_rebalance:
            self.distribute()
            self.rebalances += 1
            self._want_rebalance = False
            self._mark("rebalance")
        # bodies that crossed a splitter go to their new owner
        send, recv = self._classify()
        if recv is None:
            # More bodies are bound for some rank than it has room for: the distribution over key space changed more in
            # one step than the spare slots allow.  Everybody sees the same counts and takes the same way out: the
            # leavers go back to their slots, new splitters are chosen from the bodies as they are now, and the
            # classification is done again.
            for r in R:
                r.unclassify()
            self.distribute()
            self.rebalances += 1
            self._mark("rebalance")
            send, recv = self._classify()
            if recv is None:
                raise EngineError("the ranks' capacities do not hold the bodies even with new splitters: upload with more room per rank")
        unit = R[0].buffers.record_bytes
        self.comm.alltoall(R, "outbox", "inbox", send, unit, recv_counts=recv)
        for r, rc in zip(R, recv):
            r.receive(sum(rc))
        self._mark("migrate")
        # every rank's own tree; the roots of its fragments and its bounding box go to everybody
        built = [r.build() for r in R]
        self._mark("sort_build")
        # (one collective; the number of bodies rides along: every rank sees every rank's, and all take the same decision)
        got = self.comm.allgather_many([[b[0], b[1].reshape(-1), np.array([b[2], r.capacity], dtype=np.int64)] for b, r in zip(built, R)])
        frags = [[g[0] for g in per_rank] for per_rank in got]
        boxes = [[g[1].reshape(-1, 6) for g in per_rank] for per_rank in got]
        self.lives = [int(g[2][0]) for g in got[0]]
        fill = max(g[2][0] / g[2][1] for g in got[0])
        if self.rebalance_at is not None and sum(self.lives) > 0:
            mean = sum(self.lives) / W
            self._want_rebalance = max(self.lives) > self.rebalance_at * mean + 64 or fill > 0.9
        exported = []
        for k, r in enumerate(R):
            counts = np.array([len(f) for f in frags[k]], dtype=np.int32)
            allf = np.concatenate(frags[k]) if counts.sum() else np.zeros(0, dtype=FRAGMENT)
            exported.append((r.export(allf, counts, boxes[k]), counts))
        self._mark("export")
        # pruned fragments to those who need them, and how large each came out
        node_bytes = R[0].buffers.node_bytes
        sendn = [np.diff(ex[0][1].astype(np.int64)) for ex in exported]
        got = self.comm.allgather_many([[ex[0][0].reshape(-1), ex[0][1].astype(np.int64)] for ex in exported])
        sizes = [[g[0] for g in per_rank] for per_rank in got]
        # what every rank exports to this one follows from everybody's offsets: no exchange of counts
        recvn = [[int(got[k][src][1][r.rank + 1] - got[k][src][1][r.rank]) for src in range(W)] for k, r in enumerate(R)]
        self.comm.alltoall(R, "export", "import", sendn, node_bytes, recv_counts=recvn)
        self.last_import_nodes = [int(sum(rc)) for rc in recvn]
        self.last_migrated = [int(sum(rc)) for rc in recv]
        for k, r in enumerate(R):
            counts = exported[k][1]
            imp = []
            for src in range(W):
                nf = int(counts[src])
                m = sizes[k][src].reshape(W, nf) if nf else np.zeros((W, 0), dtype=np.int32)
                imp.append(m[r.rank])
            offs = np.concatenate([[0], np.cumsum(recvn[k])[:-1]]).astype(np.uint64)
            r.assemble(np.concatenate(imp) if len(imp) else np.zeros(0, np.int32), offs)
        self._prepared = True
        self._mark("exchange_assemble")

    def _classify(self):
        """classify() on the ranks of this process, then everybody's counts and room in one gather.  Returns (send, recv)
        with recv[k][src] = records rank k of this process receives from rank src, or (send, None) when some rank of the
        run has less room than what is bound for it."""
        R, W = self.ranks, self.world
        send = [r.classify() for r in R]
        self._mark("classify")
        got = self.comm.allgather_many([[np.asarray(s, dtype=np.int64), np.array([r.room()], dtype=np.int64)] for s, r in zip(send, R)])
        rows = [g[0] for g in got[0]]                  # rows[src][dst], the same on every rank
        room = [int(g[1][0]) for g in got[0]]
        if any(sum(int(rows[src][q]) for src in range(W)) > room[q] for q in range(W)):
            return send, None
        return send, [[int(rows[src][r.rank]) for src in range(W)] for r in R]

    def download(self):
        """(index, pos, vel, mass) of the bodies of the ranks in this process, sorted by run-wide index."""
        parts = [r.download() for r in self.ranks]
        idx = np.concatenate([p[0] for p in parts])
        order = np.argsort(idx, kind="stable")
        return tuple(np.concatenate([p[k] for p in parts])[order] for k in range(4))
