cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; \
timeout -k 10 900 python -m pytest tests/test_gpu_verify.py -m gpu -x -q -k "one_pairing" > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 12 $O/gpu_tests.log
