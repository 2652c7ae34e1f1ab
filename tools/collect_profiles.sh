#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, and the PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs, as the MI355X guide prescribes).  Outputs under gpurun_out/<round>/.
#   tools/collect_profiles.sh r02            everything
#   tools/collect_profiles.sh r02 xcd        + FETCH_SIZE pass with the XCD-chunked tile walk (option xcd_chunks=8)
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || exit 1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err || exit 1
  echo "pmc $c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err || echo "mfma pmc pass failed"
echo "pmc mfma done"
if [ "$2" = "xcd" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_FETCH_SIZE_xcd -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile --opt xcd_chunks=8 > $OUT/pmc_FETCH_SIZE_xcd.json 2> $OUT/pmc_FETCH_SIZE_xcd.err || echo "xcd pass failed"
  echo "pmc xcd done"
fi
ls -R $OUT | head -40
