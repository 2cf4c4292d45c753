# Round 4, second GPU call: the full GPU test tier; launch timelines of the mixed batch at 512 / 2048 / 4096 ops (where the fixed cost of a
# batch sits); gather-rate calibration with the MSM's locality (512 KB regions), table sizes around the Infinity Cache, and more waves.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4b && mkdir -p $O && cd $R \
&& python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/gpu_tests.log; tail -n 3 $O/gpu_tests.log; \
for n in 512 2048 4096; do rm -f $O/trace_$n.jsonl; ZKP_HIP_TRACE=$O/trace_$n.jsonl python3 tools/enqueue_time.py $n 7 > $O/enqueue_$n.txt 2>&1; python3 tools/trace_timeline.py $O/trace_$n.jsonl > $O/timeline_$n.txt; cat $O/enqueue_$n.txt; done; \
( ./build/tools/gather_calib 16 64 3072 512 ; ./build/tools/gather_calib 16 64 3072 1024 ; ./build/tools/gather_calib 64 64 3072 512 ; ./build/tools/gather_calib 64 64 3072 0 ; \
  ./build/tools/gather_calib 1 64 3072 0 ; ./build/tools/gather_calib 1 64 3072 0 128 ; ./build/tools/gather_calib 16 64 8192 512 ; ./build/tools/gather_calib 16 64 1024 512 ; ./build/tools/gather_calib 16 64 2048 512 ) > $O/gather_calib.jsonl 2>$O/err_calib.txt; cat $O/gather_calib.jsonl | cut -c1-250; ls $O
