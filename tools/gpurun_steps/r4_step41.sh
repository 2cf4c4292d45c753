cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; \
timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_stark.py tests/test_gpu_bits.py -m gpu -x -q > $O/gpu_tests_v.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests_v.log
