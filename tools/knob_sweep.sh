#!/bin/bash
# Re-check the batch schedule knobs after a kernel change (debug library: the rejected experiments live there).  Runs on the GPU box.
export SIGP_USE_DEBUG_LIB=1
OUT=${1:-gpurun_out/knobs}
mkdir -p $OUT
run() { python3 bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %7.2f fits/s  frac %.3f' % (' '.join(sys.argv[1:]), d['value'], d['roofline']['frac']))" "$@"; }
for rep in 1 2; do
  run
  run --concurrency 2
  run --opt strips_after_update=1
  run --outer 16
  run --outer 4
  run --opt first_on_panel=2
  run --opt panel_chain=1
  run --group 80 --years 80
  run --opt n64_tiles=1
  run --opt xcd_chunks=8
done
