#!/bin/bash
# A/B on ONE box: an older tree against the working tree, c4 shard replayed in both replay modes of the runtime, with replay_check.
# Prepare the older tree in the container first (it travels to the box with the snapshot; _old/ is git-ignored):
#   mkdir _old && git archive <rev> | tar -x -C _old && (cd _old && python -c 'import __graft_entry__ as g; g.build()')
# Used for profiles/r06_graph_alloc_hazard.txt 5b (<rev> = 2946ba8, the last tree with memset nodes in the captured step).
mkdir -p gpurun_out/r06
out=$PWD/gpurun_out/r06/ab_c4.txt
: > $out
one() {  # dir label packets
  (cd $1 && timeout -k 10 300 python bench.py --workload c4 --graph --graph-packets $3 --steps 20 --warmup 3 2> /tmp/ab.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); rc = d.get('replay_check') or {}
print('$2 $3:', d['ms_per_step'], 'ms  replay_check', rc.get('params_rel_l2_replay_vs_eager'), 'ok', rc.get('ok'))" >> $out; grep "timed steps" /tmp/ab.err | cut -c1-200 >> $out)
}
for rep in 1 2; do
  one _old old safe
  one . new safe
  one . new fast
  one _old old fast
done
cat $out
