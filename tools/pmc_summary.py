"""Summarise a `rocprofv3 --pmc ... --output-format csv` run per kernel.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.json]

For every ca_* kernel: launches, summed wall (End-Start) and every collected counter (sum and per launch).
If SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE are present it also derives
  clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / wall           (MI355X_MICROARCH.md 'DVFS give-back')
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
i.e. the share of SIMD-cycles, at the clock the chip actually held, in which the matrix pipe was busy.
"""
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    m = re.search(r"(ca_[a-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else "other"


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k == "other":
                continue
            e = per.setdefault(k, {"launches": 0, "wall_us": 0.0, "counters": {}})
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                e["launches"] += 1
                e["wall_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            c = r["Counter_Name"]
            e["counters"][c] = e["counters"].get(c, 0.0) + float(r["Counter_Value"])
    for k, e in per.items():
        c = e["counters"]
        e["wall_us_per_launch"] = e["wall_us"] / max(e["launches"], 1)
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            e["clock_ghz"] = cyc / (e["wall_us"] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        if "SQ_BUSY_CU_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["SQ_BUSY_CU_CYCLES"] > 0:
            e["mfma_busy_over_cu_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])
    out = json.dumps(per, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    for k, e in sorted(per.items(), key=lambda kv: -kv[1]["wall_us"]):
        extra = {x: round(e[x], 4) for x in ("clock_ghz", "mfma_busy_frac", "mfma_busy_over_cu_busy") if x in e}
        print(f"{k:40s} n={e['launches']:5d} wall/launch={e['wall_us_per_launch']:9.1f}us {extra}")


if __name__ == "__main__":
    main()
