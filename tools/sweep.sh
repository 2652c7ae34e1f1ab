#!/bin/bash
# usage: tools/sweep.sh "<bench args>" ...   -> one compact line per configuration
for a in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $a > /tmp/sweep.json 2> /tmp/sweep.err || { echo "FAILED: $a"; tail -3 /tmp/sweep.err; continue; }
  python -c "
import json
d=json.loads(open('/tmp/sweep.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('$a', '| fits/s %.1f ms/fit %.2f' % (d['value'], d['ms_per_step']), '| outer TF %.1f' % r.get('achieved',0), {k:round(v['ms_per_fit'],2) for k,v in d.get('kernels',{}).items()})"
done
