cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab13.jsonl; \
for i in 1 2 3; do \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab13.jsonl; \
ZKP_HIP_G16_SIDE2_PRIORITY=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab13.jsonl; \
done
