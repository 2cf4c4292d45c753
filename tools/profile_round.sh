cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r2v3 && mkdir -p $O && cd $R \
&& rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_under_rocprof.json 2>$O/err_kt.txt \
&& python3 tools/timeline.py $O/kt/p_kernel_trace.csv > $O/timeline.txt \
&& rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_s -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_s.txt \
&& rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_l -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_l.txt \
&& rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_f.txt \
&& rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>$O/err_w.txt \
&& python3 bench.py > $O/bench.json 2>$O/err_bench.txt && tail -n 1 $O/bench.json | cut -c1-300 && ls $O $O/pmc_s | head -30
