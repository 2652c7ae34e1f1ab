#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, and the PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, as the MI355X guide prescribes).  Outputs under gpurun_out/<round>/.
#   tools/collect_profiles.sh r03 bench     the default bench line (with the CPU protocol: ~5 min) + kernel stats of the same command
#   tools/collect_profiles.sh r03 pmc       PMC passes of the bench command (trailing update, covariance build)
#   tools/collect_profiles.sh r03 f32       kernel stats + PMC passes of one fp32 fit at configs[4]'s shape (syrk128_kernel<float>)
# then, back in the authoring container:  python tools/summarize_pmc.py r03
R=${1:-r05}
WHAT=${2:-bench}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = "bench" ]; then
  python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1       # stdout: the compact metric line; the full record:
  cp $GRAFT_REPO_ROOT/gpurun_out/bench_verbose.json $OUT/bench.json                              # (bench.py writes it there and to stderr)
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
  echo "stats done"
fi
if [ "$WHAT" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || exit 1
    echo "pmc $c done"
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err || echo "mfma pmc pass failed"
  echo "pmc mfma done"
  # LDS bank conflicts and occupancy (the chain kernels: diag_update, panel_strip, chain_link)
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_lds -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err || echo "lds pmc pass failed"
  echo "pmc lds done"
fi
if [ "$WHAT" = "f32" ]; then
  F32="python3 $GRAFT_REPO_ROOT/tools/shard_profile.py --single --dtype f32 --n 32768 --d 32 --kernel matern52 --sn 0.1 --reps 2"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32_stats -o f32 -- $F32 > $OUT/f32_stats.log 2>&1 || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/f32_pmc_$c -o f32 -- $F32 > $OUT/f32_pmc_$c.log 2>&1 || exit 1
    echo "f32 pmc $c done"
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/f32_pmc_mfma -o f32 -- $F32 > $OUT/f32_pmc_mfma.log 2>&1 || echo "f32 mfma pass failed"
fi
ls -R $OUT | head -60
