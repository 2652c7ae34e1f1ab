#!/bin/bash
# FETCH_SIZE of the trailing update with engine options A/B (one rocprofv3 --pmc pass each): tools/fetch_ab.sh "diag_tiles=0" "diag_tiles=1" ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/fetch_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for o in "$@"; do
  i=$((i+1))
  args=""
  for kv in $o; do args="$args --opt $kv"; done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p$i -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 --no-profile $args > $OUT/p$i.json 2> $OUT/p$i.err || exit 1
  python3 - "$o" $OUT/p$i/b_counter_collection.csv <<'PY'
import sys, pandas as pd
df = pd.read_csv(sys.argv[2])
df = df[df["Kernel_Name"].str.contains("syrk128_kernel<double, false", regex=False)]
print("%-40s launches %d  fetch per launch %.2f GB (x2 corrected)" % (sys.argv[1], len(df), df["Counter_Value"].mean() * 1024 * 2 / 1e9), flush=True)
PY
done
