"""Host-side pieces of the distributed tree that need no GPU."""
import numpy as np

from nbody_amd.disttree import choose_splitters


def test_splitters_follow_the_weighted_quantiles():
    rng = np.random.default_rng(1)
    keys = np.sort(rng.integers(0, 1 << 62, 40000, dtype=np.uint64))
    # four ranks hold interleaved quarters of different sizes and sample them at different rates
    parts = [keys[0:5000], keys[5000:25000], keys[25000:26000], keys[26000:]]
    samples = [p[(2 * np.arange(k) + 1) * len(p) // (2 * k)] for p, k in zip(parts, (100, 400, 50, 300))]
    split = choose_splitters(samples, [len(p) for p in parts], 4)
    assert split[0] == 0 and split[4] == np.uint64(1) << np.uint64(63) and np.all(np.diff(split.astype(np.float64)) >= 0)
    owned = np.diff(np.searchsorted(keys, split[1:4], side="left"), prepend=0, append=len(keys))
    assert np.abs(owned - 10000).max() <= 300           # balanced to the sampling resolution


def test_splitters_with_empty_and_tiny_ranks():
    split = choose_splitters([np.zeros(0, np.uint64)] * 3, [0, 0, 0], 3)
    assert list(split[1:3]) == [np.uint64(1) << np.uint64(63)] * 2
    split = choose_splitters([np.array([5, 9], np.uint64), np.zeros(0, np.uint64)], [2, 0], 2)
    assert split[1] in (5, 9)
    # equal keys never straddle a splitter: the splitter is one of the keys and everything below it goes left
    split = choose_splitters([np.full(10, 77, np.uint64)], [10], 2)
    assert split[1] == 77


# ---- the stitching: cells that span ranks, rebuilt from the fragments' roots ----------------------------------------
def morton_keys(pos, lo, width):
    """63-bit keys as the engine forms them (bit 3k = x bit k, 3k+1 = y, 3k+2 = z of the depth-21 cell index; the child
    order of octantBits, tree.go:128-132); exact for a power-of-two bound."""
    q = np.floor((pos - lo) / (width / 2 ** 21)).astype(np.uint64)
    key = np.zeros(len(pos), dtype=np.uint64)
    for bit in range(21):
        for axis in range(3):
            key |= ((q[:, axis] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + axis)
    return key


def test_spanning_cells_rebuilt_from_fragments_are_the_octrees_cells():
    """Bodies split over three key ranges; the fragments (maximal subtrees inside a range) are read off the pointer octree
    of the CPU restatement; nbe_dist_layout must put every fragment at its index in the octree's pre-order and recreate
    exactly the cells above them: same places, depths, subtree ends, masses and centres."""
    import pytest
    from nbody_amd import physics
    from nbody_amd.disttree import FRAGMENT, layout
    from oracle import oracle as O
    W, lo = 65536.0, -32768.0
    for seed, world in ((3, 3), (4, 2), (5, 7)):
        b = physics.plummer(3000, seed=seed, a=2e3)
        dump, alive = O.tree_dump(b.mass, b.pos, com_mode="true", max_depth=21)
        assert alive.all()
        keep = dump[(dump["kind"] == 0) | (dump["nchild"] >= 2)]          # leaves and branching cells, pre-order
        n = len(keep)
        keys = morton_keys(b.pos, lo, W)
        # subtree ends and parents of the kept nodes (pre-order): a node lies below a cell iff its own cell lies inside it
        # (one-child chains are not kept, so depths alone do not tell)
        depth = keep["depth"]
        end = np.zeros(n, dtype=np.int64)
        parent = np.full(n, -1, dtype=np.int64)
        stack = []

        def below(i, top):
            return (keep["kind"][top] == 1 and depth[i] > depth[top]
                    and np.all(np.abs(keep["center"][i] - keep["center"][top]) < 0.5 * keep["width"][top]))
        for i in range(n):
            while stack and not below(i, stack[-1]):
                end[stack.pop()] = i
            if stack:
                parent[i] = stack[-1]
            stack.append(i)
        for i in stack:
            end[i] = n
        # key range of every node: a leaf is its body's key, a cell the keys below its centre's cell at its depth
        first_leaf_key = np.zeros(n, dtype=np.uint64)
        for i in range(n - 1, -1, -1):
            first_leaf_key[i] = keys[keep["particle"][i]] if keep["kind"][i] == 0 else first_leaf_key[i + 1]
        kmin, kmax = [0] * n, [0] * n                                       # Python integers: the keys have 63 bits
        for i in range(n):
            k = int(first_leaf_key[i])
            size = 1 if keep["kind"][i] == 0 else 1 << (3 * (21 - int(depth[i])))
            kmin[i] = k & ~(size - 1)
            kmax[i] = kmin[i] + size - 1
        order = np.sort(keys)
        split = [0] + [int(order[len(order) * k // world]) for k in range(1, world)] + [1 << 63]
        # sums travel as 128-bit two's complement integers (low word first) in a quantum of the run: here 2^-40
        def words(vals):
            out = []
            for v in vals:
                q = int(round(float(v) * 2.0 ** 40)) % (1 << 128)
                out += [q & ((1 << 64) - 1), q >> 64]
            return out

        def values(w):
            out = []
            for k in range(4):
                q = int(w[2 * k]) | (int(w[2 * k + 1]) << 64)
                out.append((q - (1 << 128) if q >> 127 else q) / 2.0 ** 40)
            return np.array(out)
        frags, sizes, where = [], [], []
        for r in range(world):
            inside = [kmin[i] >= split[r] and kmax[i] < split[r + 1] for i in range(n)]
            for i in range(n):
                if inside[i] and (parent[i] < 0 or not inside[parent[i]]):
                    f = np.zeros(1, dtype=FRAGMENT)
                    f["key"] = first_leaf_key[i]
                    f["level"] = 255 if keep["kind"][i] == 0 else depth[i]
                    f["start"], f["nodes"] = i, end[i] - i
                    f["sum"][0] = words([keep["mass"][i]] + [keep["mass"][i] * keep["com"][i][k] for k in range(3)])
                    frags.append(f); sizes.append(end[i] - i); where.append(i)
        frags = np.concatenate(frags)
        assert all(int(a) < int(b) for a, b in zip(frags["key"][:-1], frags["key"][1:]))
        tops, dst, total = layout(frags, sizes)
        assert total == n and np.array_equal(dst, where)
        covered = np.zeros(n, dtype=bool)
        for i, k in zip(where, sizes):
            covered[i:i + k] = True
        assert sorted(tops["dst"]) == list(np.nonzero(~covered)[0])       # every node outside the fragments is a spanning cell
        for t in tops:
            i = t["dst"]
            assert keep["kind"][i] == 1 and t["level"] == depth[i] and t["skip"] == end[i]
            got = values(t["sum"])
            assert got[0] == pytest.approx(keep["mass"][i], rel=1e-13)
            assert np.abs(got[1:] / got[0] - np.array(keep["com"][i])).max() <= 1e-9
