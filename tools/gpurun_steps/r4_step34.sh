cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab9.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab9.jsonl; \
for v in sum64 sum128; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab9.jsonl; done; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab9.jsonl; \
for v in sum64 sum128; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab9.jsonl; done; tail -n 3 $O/err_ab.txt
