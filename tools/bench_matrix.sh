#!/bin/bash
# bench.py --no-cpu lines for a list of "label:args" variants -> one summary line each (value, ms/step, kernel ms)
cd $GRAFT_REPO_ROOT
O=$1; shift
mkdir -p $O
for v in "$@"; do
  L=${v%%:*}; A=${v#*:}
  python3 bench.py --no-cpu --steps 5 --warmup 2 $A > $O/$L.json 2> $O/$L.err
  python3 -c "
import json,sys
d=json.load(open('$O/$L.json'))
k={x['name'].split(' ')[0]: round(x['ms'],3) for x in d['roofline']['kernels']}
print('$L', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms', k)
"
done
