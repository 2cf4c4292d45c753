cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4z && mkdir -p $O && cd $R; rm -f $O/ab5.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
ZKP_HIP_ENCODE_CAPPED_MIN=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
ZKP_HIP_ENCODE_CAPPED_MIN=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
ZKP_HIP_ENCODE_CAPPED_MIN=0 timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab5.jsonl; \
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests.log
