"""bench.py's contract: one JSON line with the metric, roofline and CPU baseline; and the branches that only a run with
several ranks takes, rehearsed with two ranks on this GPU over gloo (small sizes, the extras left out)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_fields(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--bodies", "65536", "--steps", "3", "--warmup", "1", "--no-extra",
                        "--cpu-sample", "2048"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert d["metric"] == "body-body interactions/sec" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert abs(d["value"] - 65536 * 65535 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    ro, cb = d["roofline"], d["cpu_baseline"]
    assert ro["unit"] == "TFLOP/s" and 0 < ro["frac"] < 1 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


@pytest.mark.parametrize("workload", ["direct", "tree"])
def test_two_ranks_rehearsed_over_gloo(gpu, workload):
    env = dict(os.environ, NBODY_BENCH_REHEARSE="gloo")
    port = 29600 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bodies", "131072", "--steps", "3",
                        "--warmup", "1", "--no-extra", "--workload", workload], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["extra"]["world"]["rccl_ranks"] == 2
    assert d["extra"]["world"]["backend"] == "gloo" and set(d["extra"]["headline_sharded_phase_ms"]) >= {"advance"}
    assert d["value"] > 0 and d["ms_per_step"] > 0
