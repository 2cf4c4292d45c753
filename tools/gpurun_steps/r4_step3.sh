# Round 4, third GPU call: the HBM-resident ed25519 tables (edg.h) -- smoke, the GPU test tier, then A/B of the gather kernel's variants
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4c && mkdir -p $O && cd $R \
&& timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 4 $O/smoke.log; \
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests.log; \
for v in "" _edg41 _edg31 _edg4l _edg32; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab$v.txt | tee -a $O/ab.jsonl; done; \
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2>$O/err_bench.txt; tail -n 1 $O/bench.json | cut -c1-300; \
rm -f $O/trace.jsonl; ZKP_HIP_TRACE=$O/trace.jsonl python3 tools/enqueue_time.py 4096 7 > $O/enqueue.txt 2>&1; python3 tools/trace_timeline.py $O/trace.jsonl > $O/timeline_4096.txt; cat $O/enqueue.txt
