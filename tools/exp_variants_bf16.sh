#!/bin/bash
# Builds ablation variants of the library for the bf16 conv kernel (developer tool)
set -e
cd "$(dirname "$0")/.."
SRC=pytorch_realtime_multi-person_pose_estimation_amd/csrc
mkdir -p tools/exp
build() { # name flags...
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRTPOSE_DEV_BUILD -Iinclude -I$SRC "$@" -c $SRC/conv_mfma_bf16.hip -o tools/exp/${name}_bf.o
  objs="tools/exp/${name}_bf.o"
  for f in conv_mfma layout_ops net shufflenet decode legacy_pafprocess; do objs="$objs $SRC/build/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/lib_$name.so $objs
  rm -f tools/exp/${name}_bf.o
}
for v in "$@"; do
  case $v in
    bnob) build bnob -DRTPOSE_EXP_NO_B & ;;
    bnoa) build bnoa -DRTPOSE_EXP_NO_A & ;;
    bnostage) build bnostage -DRTPOSE_EXP_NO_STAGE & ;;
    bnoab) build bnoab -DRTPOSE_EXP_NO_A -DRTPOSE_EXP_NO_B & ;;
    bhd3) build bhd3 -DRTPOSE_EXP_HD=3 & ;;
    bhd9) build bhd9 -DRTPOSE_EXP_HD=9 & ;;
    btime) build btime -DRTPOSE_EXP_TIMELINE & ;;
    bnofill) build bnofill -DRTPOSE_EXP_NO_FILL & ;;
    bnostore) build bnostore -DRTPOSE_EXP_NO_STORE & ;;
    bnone) build bnone -DRTPOSE_EXP_NO_A -DRTPOSE_EXP_NO_B -DRTPOSE_EXP_NO_STAGE -DRTPOSE_EXP_NO_FILL -DRTPOSE_EXP_NO_STORE & ;;
  esac
done
wait
ls -la tools/exp
