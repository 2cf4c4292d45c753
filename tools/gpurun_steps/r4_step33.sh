cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab8.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab8.jsonl; \
for c in 1 2 3; do ZKP_HIP_BP_CUS=$c ZKP_HIP_BP_CUS_SHARED=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab8.jsonl; done; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab8.jsonl; \
for c in 1 2; do ZKP_HIP_BP_CUS=$c ZKP_HIP_BP_CUS_SHARED=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab8.jsonl; done; tail -n 5 $O/err_ab.txt
