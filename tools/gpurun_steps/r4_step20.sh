cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4u2 && mkdir -p $O && cd $R; \
timeout -k 10 900 python -m pytest tests/test_gpu_verify.py -m gpu -x -q -k "groth16" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests.log; \
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o vg -- python3 $R/tools/verify_g16_time.py 4096 --sweep > $O/verify_g16.json 2>$O/err_vg.txt; echo "vg rc $?"; cat $O/verify_g16.json; tail -n 5 $O/err_vg.txt
