#!/bin/bash
# Round-3 session 35: launch gaps - the forward as a hipGraph replay (RTPOSE_GRAPH=1) against plain launches
cd "$(dirname "$0")/../.."
for g in 0 1 0 1; do
  echo "RTPOSE_GRAPH=$g: $(RTPOSE_GRAPH=$g timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
