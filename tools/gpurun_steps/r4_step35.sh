cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab10.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab10.jsonl; \
ZKP_HIP_BP_PRIORITY=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab10.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab10.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab10.jsonl; \
ZKP_HIP_BP_PRIORITY=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab10.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab10.jsonl
