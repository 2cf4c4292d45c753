cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4w && mkdir -p $O && cd $R; \
timeout -k 10 900 python3 tools/verify_g16_batch_sweep.py > $O/verify_g16_batch.json 2>$O/err.txt; echo "rc $?"; cat $O/verify_g16_batch.json; tail -n 5 $O/err.txt
