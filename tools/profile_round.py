#!/usr/bin/env python3
"""One command that produces every profile the bench line's roofline numbers rest on, stamped with the git HEAD.

    python tools/profile_round.py --round 2            # in the build container: drives ONE gpurun call
    python tools/profile_round.py --on-box --round 2   # what that call executes on the GPU box
    python tools/profile_round.py --round 5 --workload sweep [--batch 5]     # (round 5) the same for another workload of
                                                       # bench.py: files profiles/rNN_<workload>[_b<batch>]_*; --no-pmc skips
                                                       # the three counter passes (kernel trace + bench line only)

On the box, for the exact bench command (`python3 bench.py --steps 5 --warmup 1 --streams 1 --profile-mode`: one
warm-up group and one timed group of 5 work items on one stream and NOTHING else -- no single-item re-run, no
block-timing forward, no per-launch events -- so every launch in the trace is a 5-item launch, the two groups hold the
same launch list, and a kernel's AverageNs is the average over one 5-item call's launches, which is what bench.py's
`roofline.avg_launch_us` measures with HIP events; the launch counts bench.py reports for ONE group are checked against
the trace: Calls == 2 x that):
  1. rocprofv3 --kernel-trace --stats                       -> profiles/rNN_rocprofv3_kernel_stats.csv
  2. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace
                                                            -> profiles/rNN_pmc_mfma_busy.json
  3. rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes: the TCC block holds one of them at a
     time, MI355X_MICROARCH.md)                             -> profiles/rNN_pmc_hbm_traffic.json
  4. the un-profiled default bench line                     -> profiles/rNN_bench_default.json
Counter passes never combine --pmc with sys/hip/hsa tracing.  The summaries carry {"git_head": ...}; bench.py reads
`roofline.traffic` from the newest rNN_pmc_hbm_traffic.json.
"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = ["python3", "bench.py", "--steps", "5", "--warmup", "1", "--streams", "1", "--profile-mode"]
LAUNCH_MIX = {"items_per_forward": 5, "workload": "generate", "model": "flux-schnell", "concepts": 4, "size": 1024,
              "groups_in_process": 2, "only_batched_launches": True, "precision": "bf16"}


def configure(workload: str, batch: int):
    """Bench command, launch mix and file tag of one (workload, items per forward) pair; the default pair keeps the
    names of rounds 1-4 (rNN_rocprofv3_kernel_stats.csv ...)."""
    global BENCH, LAUNCH_MIX
    BENCH = ["python3", "bench.py", "--workload", workload, "--batch", str(batch), "--steps", str(batch), "--warmup", "1",
             "--streams", "1", "--profile-mode"]
    LAUNCH_MIX = dict(LAUNCH_MIX, workload=workload, items_per_forward=batch, concepts=2 if workload == "encode" else 4)
    return "" if (workload, batch) == ("generate", 5) else f"{workload}{'' if batch == 5 else f'_b{batch}'}_"


def sh(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, **kw)


def on_box(rnd: int, head: str, sub: str = "", pmc: bool = True, batch: int = 5):
    tag = f"r{rnd:02d}_{sub}".rstrip("_") if sub else f"r{rnd:02d}"
    out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    env = dict(os.environ, TMPDIR="/tmp")
    os.chdir(ROOT)
    # 1. kernel trace + stats
    d = os.path.join(out, "trace")
    sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + BENCH, env=env,
       stdout=open(os.path.join(out, "trace.log"), "w"), stderr=subprocess.STDOUT, check=True)
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    lines = open(stats[0]).read().splitlines()
    # bench.py --profile-mode counted the calls of ONE group; the trace holds the warm-up group and the timed group
    trace_log = open(os.path.join(out, "trace.log")).read()
    counts = json.loads([l for l in trace_log.splitlines() if l.startswith("{")][-1])["launch_counts"]
    import csv as _csv
    calls = {}
    for r in _csv.DictReader(lines):   # (attention: the bf16 and the half-precision-q/k instantiation together)
        for key, pat in (("gemm_tile_5", "ca_gemm_pp_kernel<2, 2, false>"), ("attn", "ca_attn4_"),
                         ("heatmap_fused", "ca_heatmap_fused_kernel")):
            if pat in r["Name"]:
                calls[key] = calls.get(key, 0) + int(r["Calls"])
    mix = dict(LAUNCH_MIX, calls_per_group=counts, calls_in_trace=calls)
    for key, n in calls.items():
        if key == "heatmap_fused":
            continue
        if n != 2 * counts.get(key, -1):   # bench.py then refuses the traffic file (only_batched_launches False)
            print(f"profile_round: the trace holds {n} launches of {key}, two 5-item groups make "
                  f"{2 * counts.get(key, -1)}: some launch in the process does not have the timed shape", flush=True)
            mix["only_batched_launches"] = False
    with open(os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"), "w") as f:
        f.write(f"# git_head {head}; command: rocprofv3 --kernel-trace --stats -- {' '.join(BENCH)}; launch_mix "
                f"{json.dumps(mix)}\n")
        # (the library's kernels in full; PyTorch's -- weight initialisation, small copies -- only where they matter)
        keep = [l for i, l in enumerate(lines) if i == 0 or "ca_" in l.split(",")[0] or i < 25]
        f.write("\n".join(keep[:60]) + "\n")
    if not pmc:
        r = sh(["python3", "bench.py"] + BENCH[2:BENCH.index("--steps")] + ["--steps", str(2 * batch), "--warmup", "1"],
               env=env, capture_output=True, text=True, check=True)
        doc = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        doc["git_head"] = head
        json.dump(doc, open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
        shutil.rmtree(os.path.join(out, "trace"), ignore_errors=True)
        print("profiles written to", out)
        return
    # 2. MFMA busy
    d = os.path.join(out, "pmc_mfma")
    sh(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace",
        "--output-format", "csv", "-d", d, "--"] + BENCH, env=env,
       stdout=open(os.path.join(out, "pmc_mfma.log"), "w"), stderr=subprocess.STDOUT, check=True)
    j = os.path.join(out, f"{tag}_pmc_mfma_busy.json")
    sh([sys.executable, "tools/pmc_summary.py", d, j], check=True)
    doc = {"git_head": head, "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
           "--kernel-trace -- " + " ".join(BENCH), "launch_mix": mix, "kernels": json.load(open(j))}
    json.dump(doc, open(j, "w"), indent=1, sort_keys=True)
    # 3. HBM-side traffic, one counter per pass
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        sh(["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, "pmc_" + c),
            "--"] + BENCH, env=env, stdout=open(os.path.join(out, f"pmc_{c}.log"), "w"), stderr=subprocess.STDOUT,
           check=True)
    j = os.path.join(out, f"{tag}_pmc_hbm_traffic.json")
    sh([sys.executable, "tools/pmc_hbm_traffic.py", os.path.join(out, "pmc_FETCH_SIZE"),
        os.path.join(out, "pmc_WRITE_SIZE"), j], check=True)
    doc = json.load(open(j))
    doc["git_head"] = head
    doc["command"] = " ".join(BENCH)
    doc["launch_mix"] = mix
    n_pp = doc["kernels"].get("ca_gemm_pp_kernel<2,2>", {}).get("launches")
    if n_pp != calls.get("gemm_tile_5"):
        print(f"profile_round: the counter pass saw {n_pp} ca_gemm_pp_kernel<2,2> launches, the trace "
              f"{calls.get('gemm_tile_5')}", flush=True)
        doc["launch_mix"] = dict(mix, only_batched_launches=False)
    json.dump(doc, open(j, "w"), indent=1)
    shutil.copy(j, os.path.join(ROOT, "profiles", os.path.basename(j)))  # step 4 reads roofline.traffic from it
    # 4. the default, un-profiled bench line (groups of 5 work items, per-launch HIP-event timing on the last group)
    # (twice for the short-step workloads, the second line kept: the first un-profiled process after the profiled ones has
    # repeatedly measured 15 % slow on this pool -- 25.5 vs 22.0 ms per sweep level, round 5 -- whatever its flags; the
    # first line's value is recorded beside it)
    first = None
    if sub:
        r0 = sh(["python3", "bench.py"] + BENCH[2:BENCH.index("--steps")] + ["--steps", str(4 * batch), "--warmup", "1"],
                env=env, capture_output=True, text=True, check=True)
        first = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][-1])
    r = sh(["python3", "bench.py"] + BENCH[2:BENCH.index("--steps")] + ["--steps", str((4 if sub else 2) * batch),
                                                                       "--warmup", "1"], env=env,
           capture_output=True, text=True, check=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    doc = json.loads(line)
    doc["git_head"] = head
    if first is not None:
        doc["first_process_of_the_pair"] = {"value": first["value"], "ms_per_step": first["ms_per_step"]}
    json.dump(doc, open(os.path.join(out, f"{tag}_bench_default.json" if not sub else f"{tag}_bench.json"), "w"), indent=1)
    for big in ("trace", "pmc_mfma", "pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):  # raw traces stay on the box
        shutil.rmtree(os.path.join(out, big), ignore_errors=True)
    print("profiles written to", out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, required=True)
    ap.add_argument("--on-box", action="store_true")
    ap.add_argument("--head", default="")
    ap.add_argument("--workload", default="generate", choices=("generate", "encode", "sweep"))
    ap.add_argument("--batch", type=int, default=5)
    ap.add_argument("--no-pmc", action="store_true")
    a = ap.parse_args()
    sub = configure(a.workload, a.batch)
    if a.on_box:
        return on_box(a.round, a.head, sub, not a.no_pmc, a.batch)
    head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, text=True).strip()
    if subprocess.check_output(["git", "status", "--porcelain"], cwd=ROOT, text=True).strip():
        head += "+dirty"
    cmd = (f"python3 tools/profile_round.py --on-box --round {a.round} --head {head} --workload {a.workload} --batch {a.batch}"
           f"{' --no-pmc' if a.no_pmc else ''} > gpurun_out/profile_round_{sub or 'default'}.log 2>&1")
    rc = sh(["/usr/local/graft/bin/gpurun", "--timeout", "900", "--", cmd]).returncode
    tag = f"r{a.round:02d}_{sub}".rstrip("_") if sub else f"r{a.round:02d}"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    for f in glob.glob(os.path.join(src, f"{tag}_*")):
        shutil.copy(f, os.path.join(ROOT, "profiles", os.path.basename(f)))
        print("copied", os.path.basename(f))
    sys.exit(rc)


if __name__ == "__main__":
    main()
