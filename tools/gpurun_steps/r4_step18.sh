cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4s && mkdir -p $O && cd $R; \
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 3 $O/smoke.log; \
timeout -k 10 300 python3 tools/bench_verify.py > $O/verify_rates.json 2>$O/err_verify.txt; echo "verify rc $?"; cat $O/verify_rates.json; tail -n 3 $O/err_verify.txt; \
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests.log
