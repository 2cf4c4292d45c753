cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab6.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab6.jsonl; \
for v in g1p3 g1p4; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab6.jsonl; done; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab6.jsonl; \
for v in g1p3 g1p4; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab6.jsonl; done
