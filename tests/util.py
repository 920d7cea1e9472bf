"""Shared helpers of the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_small.npz")


def relerr(a, b):
    """Per-body relative error |a-b| / |b| of two [n,3] arrays."""
    return np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)


def f32(a):
    """Round to fp32 and back: the values an fp32 engine actually holds."""
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def load_golden():
    z = np.load(GOLDEN, allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_cases(g):
    return sorted({k.split("/")[0] for k in g})
