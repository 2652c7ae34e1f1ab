OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_traces
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in ${SIZES:-4096 8192}; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/s$n -o t -- python3 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --n $n --d $([ $n -ge 16384 ] && echo 16 || echo 8) --reps 4 > $OUT/s$n.log 2>&1 || exit 1
  f=$(find $OUT/s$n -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $f > $OUT/r04_trace_single_fit_n$n.txt 2>&1
  python3 $GRAFT_REPO_ROOT/tools/trace_seq.py $f > $OUT/r04_seq_n$n.txt 2>&1
  rm -rf $OUT/s$n
done
