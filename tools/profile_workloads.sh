set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
for wl in sweep encode; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 bench.py --workload $wl --steps 5 --warmup 1 --streams 1 --profile-mode > $O/trace_$wl.log 2>&1
  f=$(find $O/trace_$wl -name '*kernel_stats.csv' | head -n1); cp $f $O/${wl}_kernel_stats.csv
  t=$(find $O/trace_$wl -name '*kernel_trace.csv' | head -n1); python3 - "$t" $O/${wl}_trace_order.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# second half = timed group
n=len(rows); rows=rows[n//2:]
t0=int(rows[0]["Start_Timestamp"])
with open(sys.argv[2],"w") as f:
    prev_end=None
    for r in rows:
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        gap = (s-prev_end) if prev_end else 0
        f.write(f"{(s-t0)/1e3:10.1f} {((e-s)/1e3):8.1f} gap {gap/1e3:7.1f}  {r['Kernel_Name'][:90]}\n")
        prev_end=e
PY
  rm -rf $O/trace_$wl
  echo done $wl
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b1 -- python3 bench.py --batch 1 --steps 2 --warmup 1 --streams 1 --profile-mode > $O/trace_b1.log 2>&1
f=$(find $O/trace_b1 -name '*kernel_stats.csv' | head -n1); cp $f $O/b1_kernel_stats.csv; rm -rf $O/trace_b1
python3 bench.py --workload sweep --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_sweep.json 2>$O/bench_sweep.err
python3 bench.py --workload encode --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_encode.json 2>$O/bench_encode.err
python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_default.json 2>$O/bench_default.err
echo ALLDONE
