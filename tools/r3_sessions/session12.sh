#!/bin/bash
# Round-3 session 12: what each load stream of wino16_f32 costs (ablation builds: results are wrong, timing only)
cd "$(dirname "$0")/../.."
export RTPOSE_W3_16=1
export SHOW="model0.7 |model0.12|model0.21|^k=3"
VARIANTS="base:: w16nb:: w16na:: w16ns:: w16n3::" ITERS=5 tools/r3_variants.sh run
