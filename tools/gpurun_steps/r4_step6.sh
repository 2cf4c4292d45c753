# Round 4, sixth GPU call: flat step list in k_msm_gather (scalar metadata) -- smoke, GPU tests, A/B figures, kernel-alone pass
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4f && mkdir -p $O && cd $R \
&& timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 3 $O/smoke.log; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl; \
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -n 5 $O/gpu_tests.log; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>>$O/err_ab.txt | tee -a $O/ab.jsonl; \
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_s -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_s.txt; \
python3 tools/kernel_alone.py $O/pmc_s/p_kernel_trace.csv $O/kernel_alone.csv; cat $O/kernel_alone.csv | head -12; rm -f $O/*/p_agent_info.csv
