#!/bin/bash
# Developer build of the library: tools/exp/lib_dev.so, compiled with -DRTPOSE_DEV_BUILD so that the
# RTPOSE_CONV_* / RTPOSE_BF16_* environment knobs are read.  The production library
# (csrc/Makefile) ignores them.  Use:  RTPOSE_LIB_PATH=tools/exp/lib_dev.so tools/ab_env.sh VAR v1 v2
set -e
cd "$(dirname "$0")/.."
SRC=pytorch_realtime_multi-person_pose_estimation_amd/csrc
mkdir -p tools/exp
objs=""
for f in conv_mfma conv_mfma_bf16 layout_ops net shufflenet decode legacy_pafprocess; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRTPOSE_DEV_BUILD "$@" -Iinclude -I$SRC -c $SRC/$f.hip -o tools/exp/dev_$f.o &
  objs="$objs tools/exp/dev_$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/lib_dev.so $objs
rm -f $objs
ls -la tools/exp/lib_dev.so
