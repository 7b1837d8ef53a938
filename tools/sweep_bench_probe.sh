set -e
O=gpurun_out/r05l; mkdir -p $O
python tools/thin_ab.py > $O/thin_micro_head.txt 2>&1; cat $O/thin_micro_head.txt
CA_LIB_PATH=tools/ab/thin_slots4/libca.so python tools/thin_ab.py > $O/thin_micro_slots4.txt 2>&1; cat $O/thin_micro_slots4.txt
python tools/bench_ab.py --reps 3 tools/ab/thin_slots4/libca.so HEAD > $O/thin_ring_ab.txt 2>&1; cat $O/thin_ring_ab.txt
python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -x -q -k "gemm or modulation or batched" > $O/tests_k.log 2>&1 || (tail -40 $O/tests_k.log; exit 1)
tail -2 $O/tests_k.log
