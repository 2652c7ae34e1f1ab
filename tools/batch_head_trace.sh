#!/bin/bash
# Kernel sequence of the head and the tail of one lockstep step of the bench (where the update queue idles), and the schedule knobs that touch them
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_batch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-profile --steps 2 > $OUT/bench.log 2>&1 || exit 1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_seq.py $f 0 14000 > $OUT/head_seq.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/trace_seq.py $f 104000 130000 > $OUT/tail_seq.txt 2>&1
rm -rf $OUT/t
cd $GRAFT_REPO_ROOT
export SIGP_USE_DEBUG_LIB=1
for a in "" "--concurrency 2" "--opt pipeline_head=3" "--opt pipeline_head=1 --concurrency 2"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 4 $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %7.2f fits/s  frac %.3f' % ('$a', d['value'], d['roofline']['frac']))"
done
