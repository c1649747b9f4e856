#!/bin/bash
# Builds experimental variants of the library (developer tool): tools/exp/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
SRC=pytorch_realtime_multi-person_pose_estimation_amd/csrc
mkdir -p tools/exp
build() { # name flags...
  name=$1; shift
  objs=""
  for f in conv_mfma layout_ops net shufflenet decode legacy_pafprocess; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRTPOSE_DEV_BUILD -Iinclude -I$SRC "$@" -c $SRC/$f.hip -o tools/exp/${name}_$f.o &
    objs="$objs tools/exp/${name}_$f.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/lib_$name.so $objs
  rm -f $objs
}
build base
build nob -DRTPOSE_EXP_NO_B
build halfb -DRTPOSE_EXP_HALF_B_ON
build noa -DRTPOSE_EXP_NO_A
ls -la tools/exp
