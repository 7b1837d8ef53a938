"""N ranks == 1 rank on the real HIP path ("N-GPU output == 1-GPU output bit-for-bit per item, order-stable
gather").

Two FRESH child processes (bench.py's self-launch path, gloo rendezvous on 127.0.0.1, both pinned to this box's one GPU
through CA_BENCH_DEVICE -- RCCL refuses two ranks on one device) run the full-size generate / encode / sweep workloads
with the items sharded round-robin; rank 0 dumps the gathered maps, which must equal a single-process run of the same
items bit for bit and in item order.  What this does NOT cover: the `nccl` (RCCL) transport between two GPUs -- a
one-GPU box cannot host it; the device-tensor branches of the collectives run in a one-rank RCCL group in
tests/test_round2_gpu.py.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp, name, gpus, steps, extra):
    env = dict(os.environ, CA_DIST_BACKEND="gloo", CA_BENCH_DEVICE="cuda:0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    path = os.path.join(tmp, name + ".npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps),
                        "--warmup", "0", "--no-cpu-baseline", "--no-kernel-timing", "--dump-maps", path] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), np.load(path)


@pytest.mark.parametrize("workload,extra", [
    ("generate", ["--batch", "2"]),
    ("encode", ["--batch", "2"]),
    ("sweep", ["--batch", "2"]),
])
def test_two_ranks_equal_one_rank_bit_for_bit(tmp_path, workload, extra):
    """4 items: one rank runs them as groups [0,1],[2,3]; two ranks own [0,2] and [1,3] -- different groupings of the
    batched forward on top of the rank split, same maps."""
    tmp = str(tmp_path)
    one, a = _bench(tmp, "one", 1, 4, ["--workload", workload] + extra)
    two, b = _bench(tmp, "two", 2, 2, ["--workload", workload] + extra)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert a.shape == b.shape and a.shape[0] == 4
    assert np.isfinite(a).all()
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    assert not np.array_equal(a[0], a[1])          # the items differ: an order mix-up could not go unnoticed
    assert one["batched_equals_single"] is True and two["batched_equals_single"] is True
    # the N > 1 line says which group ran it: here two gloo ranks that were both pinned to device 0 -- exactly what a
    # reader of an 8-GPU line must be able to rule out ("distinct_devices")
    ev = two["collective"]
    assert "collective" not in one
    assert ev["backend"] == "gloo" and ev["ranks_seen"] == 2 and ev["devices"] == [0, 0] and ev["distinct_devices"] == 1
