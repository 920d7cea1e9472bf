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
