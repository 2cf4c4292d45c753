cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4h && mkdir -p $O && cd $R; \
for v in "" _gx1 _gx2; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip$v.so timeout -k 10 200 python3 tools/msm_rates.py 2>$O/err$v.txt | tee -a $O/msm_rates.jsonl; done
