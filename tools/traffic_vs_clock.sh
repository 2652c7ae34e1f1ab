#!/bin/bash
# Does the trailing update's clock (and fits/s) move with its L2-miss traffic?  The XCD-chunked tile walk (debug library, xcd_chunks = P: PxP patches
# of tiles per XCD) against the default column-major walk: bench value + roofline, sclk / power sampled by rocm-smi during the run, FETCH_SIZE by a
# rocprofv3 --pmc pass of the same command.  Runs on the GPU box: tools/traffic_vs_clock.sh r04   ->  gpurun_out/r04_traffic/
export SIGP_USE_DEBUG_LIB=1
R=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R}_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for P in 0 4 8 16; do
  opt=""; [ $P -gt 0 ] && opt="--opt xcd_chunks=$P"
  bash $GRAFT_REPO_ROOT/tools/clock_watch.sh $OUT/clock_P${P}_$rep.txt python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 1 $opt > $OUT/bench_P${P}_$rep.json 2> $OUT/bench_P${P}_$rep.err || echo "bench P=$P failed"
  echo "bench P=$P rep $rep done"
done
done
for P in 0 4 8 16; do
  opt=""; [ $P -gt 0 ] && opt="--opt xcd_chunks=$P"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_P$P -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile $opt > $OUT/pmc_P$P.json 2> $OUT/pmc_P$P.err || echo "pmc P=$P failed"
  echo "pmc P=$P done"
done
python3 $GRAFT_REPO_ROOT/tools/traffic_vs_clock_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
