cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4t && mkdir -p $O && cd $R; \
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests.log; \
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o vg -- python3 $R/tools/verify_g16_time.py 4096 > $O/prof_run.log 2>&1; echo "prof rc $?"; ls $O/prof | head
