cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab4.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab4.jsonl; \
for v in enc5 enc5l sum5; do ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_$v.so timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab4.jsonl; done; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab4.jsonl; \
ZKP_HIP_LIB=$R/libzkp_amd/lib/libzkp_hip_enc5l.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py -m gpu -x -q > $O/gpu_tests_enc5l.log 2>&1; echo "pytest rc $?"; tail -n 3 $O/gpu_tests_enc5l.log
