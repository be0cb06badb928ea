#!/bin/bash
# cost of the in-graph event pairs: c2 / c4 replayed with and without the probes; then the default (auto) command
for W in c2 c4; do
  for PR in "--no-probe" ""; do
    echo "== $W --graph $PR"
    python3 bench.py --workload $W --graph --steps 10 --warmup 3 --no-secondary --no-cpu-baseline $PR > /tmp/gp.json 2> /tmp/gp.log
    grep -a "timed steps\|probe " /tmp/gp.log | cut -c1-230
    grep -aq "timed steps" /tmp/gp.log || tail -8 /tmp/gp.log
    python3 -c "
import json; d=json.loads(open('/tmp/gp.json').read().strip().splitlines()[-1]); r=d.get('roofline'); print('roofline:', None if r is None else {k: r[k] for k in ('achieved','frac','launches','avg_launch_us','sample')}); print('hbm:', [(h['kernel'][:40], h['avg_launch_us'], h['frac']) for h in (d.get('roofline_hbm') or [])])"
  done
done
