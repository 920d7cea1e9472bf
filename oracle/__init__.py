"""CPU oracle for the n-body hot path.  Test infrastructure only: imported by tests/, the smoke test
and the benchmark's CPU-baseline leg, never by the shipped package."""
