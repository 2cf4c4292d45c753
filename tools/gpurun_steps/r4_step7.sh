cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r4g && mkdir -p $O && cd $R \
&& ./build/tools/g1_add_rate > $O/g1_add_rate_cold.jsonl 2>&1; ./build/tools/g1_add_rate sustain 4 > $O/g1_sustain.jsonl 2>&1; ./build/tools/g1_add_rate >> $O/g1_add_rate_hot.jsonl 2>&1; \
cat $O/g1_sustain.jsonl; grep '9x29' $O/g1_add_rate_cold.jsonl | grep 'per lane'; grep '9x29' $O/g1_add_rate_hot.jsonl | grep 'per lane'; \
timeout -k 10 200 python3 tools/ab_steps.py 11 2>$O/err_ab.txt | tee -a $O/ab.jsonl
