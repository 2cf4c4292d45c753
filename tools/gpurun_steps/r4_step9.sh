cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4i && mkdir -p $O && cd $R; \
for n in 512 2048 4096; do rm -f $O/trace_$n.jsonl; ZKP_HIP_TRACE=$O/trace_$n.jsonl python3 tools/enqueue_time.py $n 7 > $O/enqueue_$n.txt 2>&1; python3 tools/trace_timeline.py $O/trace_$n.jsonl > $O/timeline_$n.txt; cat $O/enqueue_$n.txt; done
