cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4p && mkdir -p $O && cd $R; \
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 1 $O/smoke.log; \
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_mixed.py tests/test_gpu_full_size.py -x -q > $O/gpu_tests_g16.log 2>&1; echo "pytest rc $?"; tail -n 3 $O/gpu_tests_g16.log; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl; \
timeout -k 10 200 python3 tools/msm_rates.py 2>>$O/err_ab.txt | tee -a $O/msm_rates.jsonl; \
./build/tools/g1_add_rate loose > $O/loose.jsonl 2>&1; cat $O/loose.jsonl
