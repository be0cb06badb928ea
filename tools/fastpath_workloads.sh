#!/bin/bash
# every benchmark workload replayed through the runtime's prepared-packet mode (--graph-packets fast), with replay_check and the node census
mkdir -p gpurun_out/r06
out=$PWD/gpurun_out/r06/fastpath_workloads.txt
: > $out
for w in c0 c1 c3 c4 c2p; do
  timeout -k 10 400 python bench.py --workload $w --graph --graph-packets fast --steps 10 --warmup 3 --no-secondary --no-cpu-baseline 2> /tmp/fw.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); rc = d.get('replay_check') or {}
print('$w fast:', d['ms_per_step'], 'ms  replay_check', rc.get('params_rel_l2_replay_vs_eager'), 'ok', rc.get('ok'), ' nodes', d.get('graph_nodes'))" >> $out || { echo "$w FAILED" >> $out; tail -5 /tmp/fw.err >> $out; }
done
cat $out
