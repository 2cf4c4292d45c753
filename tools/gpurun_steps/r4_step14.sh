cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4n && mkdir -p $O && cd $R; \
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests.log; \
bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile rc $?"; tail -n 3 $O/profile_round.log | cut -c1-300
