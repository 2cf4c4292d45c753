cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; \
timeout -k 10 420 python3 tools/soak.py 240 > $O/soak.txt 2>&1; echo "soak rc $?"; tail -n 3 $O/soak.txt; \
ZKP_HIP_G16_BATCH_VERIFY_MIN=1 timeout -k 10 300 python3 tools/soak.py 120 > $O/soak_batch_check.txt 2>&1; echo "soak (batch check forced) rc $?"; tail -n 3 $O/soak_batch_check.txt
