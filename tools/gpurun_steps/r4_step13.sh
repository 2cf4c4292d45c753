cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4m && mkdir -p $O && cd $R; \
run() { echo "== $*" | tee -a $O/rounds.jsonl; env "$@" timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err.txt | tee -a $O/rounds.jsonl; }; \
run A=1; run ZKP_HIP_G16_ROUNDS=2; run ZKP_HIP_G16_ROUNDS=3; run ZKP_HIP_G16_ROUNDS=4; run ZKP_HIP_BP_FILL=50; run ZKP_HIP_G16_FILL=67; run A=2
