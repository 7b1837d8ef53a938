"""`python bench.py --gpus N` must start its own ranks (the driver's SCALE command has no outer launcher).

Runs the real self-launch path of bench.py -- parent spawns N children through torch.distributed.run, ranks
rendezvous on 127.0.0.1, shard the items, gather, rank 0 prints ONE JSON line, the parent relays it and exits
with the children's code -- on the CPU with gloo and --stub-workload (constant tensors instead of GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=300):
    env = dict(os.environ, CA_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_json_line():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--stub-workload"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["gathered_in_item_order"] is True
    assert out["data"] == "stub" and out["metric"].startswith("STUB")
    # the record that lets a reader of the line verify the group: an all_reduce(sum) of 1 over the ranks, one entry per rank
    ev = out["collective_evidence"]
    assert ev["backend"] == "gloo" and ev["ranks_seen"] == 2 and len(ev["devices"]) == 2


def test_self_launch_propagates_child_failure():
    # a rank that dies AFTER the launch and the rendezvous must make the parent exit non-zero (the flag is valid, so
    # the parent's own argparse passes and the children really start)
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-workload", "--stub-fail-rank", "1"])
    assert r.returncode != 0
    assert "rank 1 fails on request" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no result line from a failed job


import pytest  # noqa: E402


@pytest.mark.parametrize("workload,collective", [("encode", "all_gather"), ("sweep", "all_reduce")])
def test_self_launch_encode_and_sweep_workloads(workload, collective):
    """The N-rank entries of BASELINE.json configs[3] (image-sharded encode, one all_gather) and configs[4]
    (level-sharded sweep, one all_reduce(sum)): launcher, sharding, collective and the JSON line, with stub items."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--stub-workload", "--workload", workload])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["workload"] == workload and out["collective"] == collective
    assert out["gathered_in_item_order"] is True


def test_world_size_mismatch_is_rejected():
    r = _run(["--gpus", "1", "--stub-workload"], {"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                  "MASTER_PORT": "1"}, timeout=120)
    assert r.returncode != 0


def test_a_group_that_is_not_what_gpus_asked_for_ends_the_run_non_zero():
    """VERDICT r04 #7a: with the RCCL backend, ranks_seen != --gpus or distinct_devices != --gpus (all ranks on one GPU; a
    group smaller than asked) must not produce a result line.  The record is overridden through --stub-evidence, the
    check is the one the real path runs (distributed.check_collective_evidence)."""
    ok = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-workload", "--stub-evidence", "2,2"])
    assert ok.returncode == 0 and [l for l in ok.stdout.splitlines() if l.startswith("{")], ok.stderr[-1500:]
    for bad in ("2,1", "1,2"):
        r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-workload", "--stub-evidence", bad])
        assert r.returncode != 0, bad
        assert "distinct devices" in (r.stderr + r.stdout)
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_check_collective_evidence_rules():
    from conceptattention_amd import distributed as D
    rccl = {"backend": "rccl (torch.distributed 'nccl')", "ranks_seen": 8, "distinct_devices": 8}
    D.check_collective_evidence(rccl, 8)
    D.check_collective_evidence(None, 1)
    D.check_collective_evidence({"backend": "gloo", "ranks_seen": 2, "distinct_devices": 1}, 2)      # gloo rehearsal
    D.check_collective_evidence(dict(rccl, distinct_devices=1), 8, rehearsal=True)                   # CA_BENCH_DEVICE
    with pytest.raises(SystemExit):
        D.check_collective_evidence(dict(rccl, distinct_devices=7), 8)
    with pytest.raises(SystemExit):
        D.check_collective_evidence(dict(rccl, ranks_seen=4), 8)
    # device masking per rank (every local index 0) without bus ids is not conclusive: no exit; with bus ids it is
    D.check_collective_evidence(dict(rccl, distinct_devices=1, devices=[0] * 8, pci_bus_ids=[-1] * 8), 8)
    with pytest.raises(SystemExit):
        D.check_collective_evidence(dict(rccl, distinct_devices=1, devices=[0] * 8, pci_bus_ids=[5] * 8), 8)
