cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl; \
for v in edgp1 edgp2 edgp3; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab.jsonl; done; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab.jsonl
