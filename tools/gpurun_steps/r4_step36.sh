cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab11.jsonl; \
for i in 1 2 3; do \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab11.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab11.jsonl; \
done; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=2 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab11.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 ZKP_HIP_G16_SIDE_PRIORITY=1 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab11.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 ZKP_HIP_STARK_PRIORITY=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab11.jsonl
