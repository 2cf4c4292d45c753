cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab12.jsonl; \
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 3 $O/smoke.log; \
timeout -k 10 600 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_scheduler.py tests/test_gpu_full_size.py -m gpu -x -q > $O/gpu_tests_fused.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests_fused.log; \
for i in 1 2 3; do \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab12.jsonl; \
ZKP_HIP_EDG_FUSED=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab12.jsonl; \
done
