# Round-4 profile set of the default bench command (run on the GPU box through gpurun: `bash tools/profile_round.sh`).
# kt:    kernel trace + stats of the overlapped run (B = python3 bench.py --steps N --warmup 1 --no-cpu-baseline --no-extra-legs)
# pmc_*: separate counter passes (they serialise the dispatches: every kernel alone on the GPU) -- SQ issue/wait, LDS, FETCH_SIZE, WRITE_SIZE, read requests
# then the library's own launch trace (ZKP_HIP_TRACE: a timeline without the profiler's per-dispatch host cost), the verification sweeps (row N2) and the
# default bench line (which reads the kernel_alone / traffic files this very run has just produced: they are copied into profiles/ on the box first).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4v && mkdir -p $O && cd $R \
&& rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2>$O/err_kt.txt \
&& python3 tools/timeline.py $O/kt/p_kernel_trace.csv > $O/timeline_rocprof.txt \
&& rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_s -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_s.txt \
&& rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_l -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_l.txt \
&& rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_f.txt \
&& rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_w.txt \
&& rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/pmc_r -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_r.txt \
&& python3 tools/kernel_alone.py $O/pmc_s/p_kernel_trace.csv $O/kernel_alone.csv \
&& python3 tools/traffic_json.py $O/traffic.json "k_msm_gather<G1Msm>" $O/pmc_f/p_counter_collection.csv $O/pmc_w/p_counter_collection.csv "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum passes of 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs' (tools/profile_round.sh), averaged over the G1 MSM launches of the run" 64 $O/pmc_r/p_counter_collection.csv > /dev/null \
&& python3 tools/traffic_json.py $O/traffic_g2.json "k_msm_gather<G2Msm>" $O/pmc_f/p_counter_collection.csv $O/pmc_w/p_counter_collection.csv "same passes, the G2 MSM launches" 128 $O/pmc_r/p_counter_collection.csv > /dev/null \
&& python3 tools/traffic_json.py $O/traffic_ed.json "k_msm_gather<EdGather" $O/pmc_f/p_counter_collection.csv $O/pmc_w/p_counter_collection.csv "same passes, the ed25519 MSM launches" 128 $O/pmc_r/p_counter_collection.csv > /dev/null \
&& python3 tools/summarize_profile.py $O/summary.md $O/kt/p_kernel_stats.csv $O/pmc_s/p_counter_collection.csv $O/pmc_l/p_counter_collection.csv $O/pmc_f/p_counter_collection.csv $O/pmc_w/p_counter_collection.csv $O/pmc_r/p_counter_collection.csv \
&& rm -f $O/trace.jsonl && ZKP_HIP_TRACE=$O/trace.jsonl python3 tools/enqueue_time.py 4096 7 > $O/enqueue_traced.txt 2>&1 \
&& python3 tools/trace_timeline.py $O/trace.jsonl > $O/timeline_trace.txt \
&& python3 tools/enqueue_time.py 4096 21 > $O/enqueue.txt 2>&1 \
&& cp $O/kernel_alone.csv $R/profiles/r04_kernel_alone.csv && cp $O/kernel_alone.meta.json $R/profiles/r04_kernel_alone.meta.json && cp $O/traffic.json $R/profiles/r04_traffic.json \
&& python3 tools/verify_g16_time.py 4096 --sweep > $O/verify_g16.json 2>$O/err_vg.txt \
&& python3 tools/verify_g16_batch_sweep.py > $O/verify_g16_batch.json 2>$O/err_vgb.txt \
&& python3 tools/bench_verify.py > $O/verify_rates.json 2>$O/err_vr.txt \
&& python3 bench.py > $O/bench.json 2>$O/err_bench.txt && tail -n 1 $O/bench.json | cut -c1-300 && rm -f $O/*/p_agent_info.csv $O/pmc_*/p_kernel_trace.csv $O/kt/p_kernel_trace.csv && ls $O $O/pmc_s | head -40
