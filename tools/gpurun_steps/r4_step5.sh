# Round 4, fifth GPU call: GPU test tier; chunk-count sweep of the gather MSM; kernel stats of the default bench under rocprofv3
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4e && mkdir -p $O && cd $R \
&& timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests.log; \
timeout -k 10 400 python3 tools/edg_chunks.py > $O/edg_chunks.jsonl 2>$O/err_chunks.txt; cat $O/edg_chunks.jsonl; \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2>$O/err_kt.txt; \
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_s -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_s.txt; \
python3 tools/kernel_alone.py $O/pmc_s/p_kernel_trace.csv $O/kernel_alone.csv; cat $O/kernel_alone.csv | head -12; rm -f $O/*/p_agent_info.csv $O/kt/p_kernel_trace.csv
