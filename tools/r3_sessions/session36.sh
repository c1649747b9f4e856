#!/bin/bash
# Round-3 session 36: hipGraph replay at batch 1 (latency) and batch 32 (bench), alternating
cd "$(dirname "$0")/../.."
for g in 0 1 0 1; do
  echo "RTPOSE_GRAPH=$g: $(RTPOSE_GRAPH=$g timeout 300 python tools/latency_b1.py 2>&1 | grep -E '^fp32|^bf16 ' | tr '\n' ' ')"
done
for g in 1 0 1 0; do
  echo "RTPOSE_GRAPH=$g: $(RTPOSE_GRAPH=$g timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
