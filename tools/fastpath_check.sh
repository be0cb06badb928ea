#!/bin/bash
# Round 6: memset nodes and the runtime's prepared-packet replay (profiles/r06_graph_alloc_hazard.txt section 5).
#   1. the memset-free step of the package on the prepared-packet path, every BETWEEN variant of the reproducer: losses AND parameters
#   2. the memset nodes put back (STEP_MEMSET): what they break on that path, and that the node-by-node mode is exact with them
mkdir -p gpurun_out/r06
out=gpurun_out/r06/fastpath_check.txt
: > $out
run() { env "$@" timeout -k 10 200 python tools/graph_alloc_hazard.py 2>&1 | grep -v amdgpu.ids | cut -c1-300 >> $out; echo >> $out; }
echo "## 1. memset-free step, prepared packets ON" >> $out
for b in none tiny tiny_sync clone pre fill:0.001 fill:0.05 inputs; do run NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=$b; done
echo "## 2. memset nodes put back, prepared packets ON" >> $out
run NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=tiny STEP_MEMSET=loss

run NPVP_GRAPH_PACKET_CAPTURE=1 BETWEEN=inputs STEP_MEMSET=loss
echo "## 3. memset nodes put back, node-by-node replay (the package default)" >> $out
run BETWEEN=tiny STEP_MEMSET=loss
run BETWEEN=inputs STEP_MEMSET=loss
cat $out
