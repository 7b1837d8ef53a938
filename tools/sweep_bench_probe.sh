set -e
O=gpurun_out/r05m; mkdir -p $O
for i in 1 2; do
for mode in "--no-kernel-timing" ""; do
python3 bench.py --workload sweep --steps 10 --warmup 1 --no-cpu-baseline --no-solo-check --no-block-timing $mode > $O/s.json 2>>$O/err.txt
python3 -c "
import json; d=json.load(open('$O/s.json')); print('sweep steps 10 [$mode]', round(d['ms_per_step'],2), {k: round(v,4) for k,v in d['timed_region'].items() if k!='note'})"
done
done
for mode in "--no-kernel-timing" ""; do
python3 bench.py --workload encode --steps 10 --warmup 1 --no-cpu-baseline --no-solo-check --no-block-timing $mode > $O/s.json 2>>$O/err.txt
python3 -c "
import json; d=json.load(open('$O/s.json')); print('encode steps 10 [$mode]', round(d['ms_per_step'],2), {k: round(v,4) for k,v in d['timed_region'].items() if k!='note'})"
done
