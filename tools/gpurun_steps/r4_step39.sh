cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ram.jsonl; \
for i in 1 2; do \
timeout -k 10 200 python3 tools/range_after_mixed.py 2>>$O/err_ram.txt | tee -a $O/ram.jsonl; \
ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1 timeout -k 10 200 python3 tools/range_after_mixed.py 2>>$O/err_ram.txt | tee -a $O/ram.jsonl; \
done; tail -n 3 $O/err_ram.txt
