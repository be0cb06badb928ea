#!/bin/bash
# c2 replayed: the prepared-packet replay mode (fast) against node-by-node (safe), both with replay_check
mkdir -p gpurun_out/r06
out=$PWD/gpurun_out/r06/ab_c2.txt
: > $out
one() {  # label env packets
  env $2 timeout -k 10 400 python bench.py --workload c2 --graph --graph-packets $3 --steps 10 --warmup 3 2> /tmp/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); rc = d.get('replay_check') or {}
print('$1 $3:', d['ms_per_step'], 'ms  replay_check', rc.get('params_rel_l2_replay_vs_eager'), 'ok', rc.get('ok'))" >> $out; grep "timed steps" /tmp/ab.err | cut -c1-220 >> $out
}
one - X=1 fast
one - X=1 safe
one - X=1 safe
one - X=1 fast
cat $out
