# Round 4, first GPU call: the whole GPU test tier, the default bench line (with the batch-size sweep), the counter list of this
# box's rocprofv3, and the calibration of the HBM counters for per-lane 64 / 128-byte gathers (tools/gather_calib.hip).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4a && mkdir -p $O && cd $R \
&& python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/gpu_tests.log; tail -n 3 $O/gpu_tests.log \
&& python3 bench.py > $O/bench.json 2>$O/err_bench.txt; tail -n 1 $O/bench.json | cut -c1-400 \
&& (rocprofv3 -L > $O/counters.txt 2>&1 || true) \
&& ./build/tools/gather_calib 16 64 > $O/gather_calib.jsonl 2>$O/err_calib.txt && cat $O/gather_calib.jsonl \
&& rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_f -o p -- ./build/tools/gather_calib 16 64 > /dev/null 2>$O/err_cal_f.txt \
&& rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/cal_r -o p -- ./build/tools/gather_calib 16 64 > /dev/null 2>$O/err_cal_r.txt \
&& rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/cal_h -o p -- ./build/tools/gather_calib 16 64 > /dev/null 2>$O/err_cal_h.txt; \
rm -f $O/*/p_agent_info.csv; ls $O
