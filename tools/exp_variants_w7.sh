#!/bin/bash
# What each load stream / pipeline stage of the F(4,7) kernel (csrc/conv_wino7.hip) costs: timing-only
# variants (results are WRONG) with one of them dropped, plus the weight prefetch distance.
#   tools/exp_variants_w7.sh build     (here, no GPU needed)      tools/exp_variants_w7.sh run   (on the GPU box)
cd "$(dirname "$0")/.."
VARIANTS="${VARIANTS:-base: noB:-DRTPOSE_EXP_NO_B noA:-DRTPOSE_EXP_NO_A noT:-DRTPOSE_EXP_NO_STAGE noAB:-DRTPOSE_EXP_NO_A,-DRTPOSE_EXP_NO_B noABT:-DRTPOSE_EXP_NO_A,-DRTPOSE_EXP_NO_B,-DRTPOSE_EXP_NO_STAGE pf2:-DRTPOSE_EXP_W7_PF=2 pf4:-DRTPOSE_EXP_W7_PF=4 tvalu:-DRTPOSE_EXP_W7_TMASK=1 tload:-DRTPOSE_EXP_W7_TMASK=2}"
for v in $VARIANTS; do
  name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
  if [ "$1" = "build" ]; then
    ONLY=${ONLY:-conv_wino7} OUT=tools/exp/lib_w7_$name.so tools/build_dev.sh $flags > /dev/null || echo "build of $name failed"
  else
    echo "=== $name"
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_w7_$name.so python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "${SHOW:-model2_1.0|model2_1.2|^k=7|sum of}"
  fi
done
