set -e
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || (tail -30 $O/tests.log; exit 1)
tail -3 $O/tests.log
python tools/env_ab.py --reps 3 "CA_GEMM_QUEUE=0" "CA_GEMM_QUEUE=1" > $O/queue_ab.txt 2>&1
cat $O/queue_ab.txt
python tools/env_ab.py --reps 2 --bench "--batch 1 --steps 4 --warmup 1" "CA_GEMM_QUEUE=0" "CA_GEMM_QUEUE=1" > $O/queue_ab_b1.txt 2>&1
cat $O/queue_ab_b1.txt
for n in 20; do
python3 bench.py --workload sweep --steps $n --warmup 1 --no-cpu-baseline --no-kernel-timing --no-solo-check --no-block-timing > $O/s${n}.json 2>>$O/err.txt
python3 -c "
import json; d=json.load(open('$O/s$n.json')); print('sweep steps $n', d['ms_per_step'], d['timed_region'])"
done
python3 bench.py --workload encode --steps 10 --warmup 1 --no-cpu-baseline > $O/encode.json 2>>$O/err.txt
python3 -c "
import json; d=json.load(open('$O/encode.json')); print('encode', d['ms_per_step'], d['timed_region'])"
