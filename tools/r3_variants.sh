#!/bin/bash
# Round-3 A/B variants of the Winograd kernels (developer builds, tools/build_dev.sh):
#   tools/r3_variants.sh build   (here, no GPU)      tools/r3_variants.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-base:: swold:conv_wino:-DRTPOSE_EXP_W3_SWOLD cgold:conv_wino7:-DRTPOSE_EXP_W7_CGMAJOR=0 tl3:conv_wino:-DRTPOSE_EXP_TIMELINE3}"
for v in $VARIANTS; do
  name=${v%%:*}; rest=${v#*:}; only=${rest%%:*}; flags=$(echo ${rest#*:} | tr ',' ' ')
  if [ "$1" = "build" ]; then
    if [ -z "$only" ]; then OUT=tools/exp/lib_r3_$name.so tools/build_dev.sh > /dev/null || echo "build of $name failed"
    else ONLY=$only OUT=tools/exp/lib_r3_$name.so tools/build_dev.sh $flags > /dev/null || echo "build of $name failed"; fi
  else
    echo "=== $name"
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$name.so python tools/profile_layers.py 32 368 368 ${ITERS:-5} fp32 2>&1 | grep -E "${SHOW:-model0.2 |model0.7 |model0.12|model0.21|model1_1.0|model2_1.0|model2_1.2|^k=|sum of}"
  fi
done
