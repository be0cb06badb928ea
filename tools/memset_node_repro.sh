#!/bin/bash
# the stand-alone reproducer in its variants -> gpurun_out/r06/memset_node_repro.txt (profiles/r06_graph_alloc_hazard.txt 5f)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/memset_node_repro.txt
: > $out
run() { env "$@" timeout -k 10 120 python tools/graph_memset_node_repro.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3} | cut -c1-330 >> $out; }
for b in none tiny memset:4 memset:61440 memset:1048576 memset:17179869184; do TAIL=1 run BETWEEN=$b; done
TAIL=3 run BETWEEN=memset:1048576 VERBOSE=1 REPLAYS=3
TAIL=3 run BETWEEN=memset:1048576 CHAINS=8 REPLAYS=3
for b in memset:1048576 memset:17179869184; do TAIL=1 run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 BETWEEN=$b; done
TAIL=1 run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 BETWEEN=memset:1048576 CHAINS=8
cat $out
