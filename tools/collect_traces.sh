#!/bin/bash
# Per-path kernel-trace summaries (tools/trace_summary.py) of the current build: batch group, single fits n = 4096 / 16384 (fp64), fp32 n = 32768.
# Runs on the GPU box; outputs gpurun_out/<round>_traces/*.txt (copy the ones to be judged into profiles/).
R=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R}_traces
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 "$@" > $OUT/$name.log 2>&1 || return 1
  if [ "$name" = "batch_group" ]; then
    python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $(find $OUT/$name -name "*kernel_trace.csv" | head -1) kbuild --critical-path $OUT/${R}_critical_path.json > $OUT/${R}_trace_$name.txt 2>&1
  else
    python3 $GRAFT_REPO_ROOT/tools/trace_summary.py $(find $OUT/$name -name "*kernel_trace.csv" | head -1) > $OUT/${R}_trace_$name.txt 2>&1
  fi
  rm -rf $OUT/$name
  echo "$name done"
}
run batch_group $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-profile --steps 2 || exit 1
run single_fit_n4096 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --n 4096 --d 8 --reps 5 || exit 1
run single_fit_n8192 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --n 8192 --d 8 --reps 4 || exit 1
run single_fit_n16384 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --n 16384 --d 16 --reps 3 || exit 1
run fp32_n32768 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --dtype f32 --n 32768 --d 32 --kernel matern52 --sn 0.1 --reps 2 || exit 1
ls $OUT
