"""Build profiles/rNN_pmc_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py \\
        --steps 5 --warmup 1 --streams 1 --profile-mode
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_WRITE_SIZE -- (same command)
    python tools/pmc_hbm_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE out.json

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB; on gfx950 FETCH_SIZE reports half
of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section), checked here on ca_ln_modulate_kernel whose
algorithmic read is 26.8 MB per launch."""
import csv, glob, json, os, re, sys

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(ca_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    t = (m.group(2) or "").replace(" ", "").replace(",false", "").replace(",true", ",fp8")
    return m.group(1) + t


def collect(d, counter):
    per = {}
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            e = per.setdefault(k, [0, 0.0])
            if (f, r["Dispatch_Id"]) not in seen:
                seen.add((f, r["Dispatch_Id"]))
                e[0] += 1
            e[1] += float(r["Counter_Value"])
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"method": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace --output-format csv -- "
                 "the bench command tools/profile_round.py records next to this text as `command`; per-launch averages over the "
                 "launches `launch_mix` describes; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read "
                 "stream, MI355X_MICROARCH.md 'HBM'; calibrated on ca_ln_modulate_kernel: algorithmic read 26.8 MB)",
       "note": "FETCH_SIZE counts the L2's fabric-side requests, Infinity-Cache hits included, so this is an upper "
               "bound on HBM bytes",
       "kernels": {}}
for k in sorted(set(fetch) & set(write)):
    nf, f = fetch[k]
    nw, w = write[k]
    out["kernels"][k] = {"launches": nf, "FETCH_SIZE_KB_per_launch": f / nf, "WRITE_SIZE_KB_per_launch": w / nw,
                         "bytes_per_launch": (2 * f / nf + w / nw) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:36s} n={v['launches']:5d}  {v['bytes_per_launch']/1e6:9.1f} MB/launch")
