cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4l && mkdir -p $O && cd $R; \
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -n 2 $O/smoke.log; \
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_mixed.py -x -q > $O/gpu_tests_g16.log 2>&1; echo "pytest rc $?"; tail -n 4 $O/gpu_tests_g16.log; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl; \
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_s -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_s.txt; \
python3 tools/kernel_alone.py $O/pmc_s/p_kernel_trace.csv $O/kernel_alone.csv; head -12 $O/kernel_alone.csv; \
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_l -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_l.txt; \
python3 - <<'PY'
import csv,collections,os
O=os.environ.get("O") or os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r4l")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(O+"/pmc_l/p_counter_collection.csv")):
    if "qap" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:20]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items(): print(k, dict(v))
PY
rm -f $O/*/p_agent_info.csv
