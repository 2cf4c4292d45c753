cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4k && mkdir -p $O && cd $R; \
run() { echo "== $*" | tee -a $O/prio.jsonl; env "$@" timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err.txt | tee -a $O/prio.jsonl; }; \
run A=1; run ZKP_HIP_BP_PRIORITY=0; run ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_G16_SIDE2_PRIORITY=1; run ZKP_HIP_BP_PRIORITY=0 ZKP_HIP_STARK_PRIORITY=2; run ZKP_HIP_LAUNCH_ORDER=0; run ZKP_HIP_LAUNCH_ORDER=2; run A=2; \
python3 tools/bench_stark.py 16384 | tail -n 2
