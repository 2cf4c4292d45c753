cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4j && mkdir -p $O && cd $R; \
timeout -k 10 600 python -m pytest tests/test_gpu_stark.py tests/test_gpu_mixed.py tests/test_gpu_scheduler.py -x -q > $O/gpu_tests_stark.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests_stark.log; \
python3 tools/bench_stark.py > $O/bench_stark.txt 2>&1; cat $O/bench_stark.txt; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl; \
for n in 512 2048 4096; do rm -f $O/trace_$n.jsonl; ZKP_HIP_TRACE=$O/trace_$n.jsonl python3 tools/enqueue_time.py $n 7 > $O/enqueue_$n.txt 2>&1; python3 tools/trace_timeline.py $O/trace_$n.jsonl > $O/timeline_$n.txt; cat $O/enqueue_$n.txt; done
