#!/bin/bash
# Developer build of the library: tools/exp/lib_dev.so (or $OUT), compiled with -DRTPOSE_DEV_BUILD so that the
# RTPOSE_CONV_* / RTPOSE_BF16_* environment knobs are read and the RTPOSE_EXP_* ablation macros
# (csrc/conv_exp.h) are accepted.  The production library (csrc/Makefile) ignores / rejects them.
#   tools/build_dev.sh [-DRTPOSE_EXP_...]            all sources with the flags
#   ONLY=conv_wino7 OUT=tools/exp/lib_x.so tools/build_dev.sh -DRTPOSE_EXP_NO_B
#                                                    flags on ONE source, the rest from the cached plain dev objects
# Use:  RTPOSE_LIB_PATH=tools/exp/lib_dev.so tools/ab_env.sh VAR v1 v2
set -e
cd "$(dirname "$0")/.."
SRC=pytorch_realtime_multi-person_pose_estimation_amd/csrc
OUT=${OUT:-tools/exp/lib_dev.so}
mkdir -p tools/exp/objs
FILES=$(sed -n 's/^SRCS := //p' $SRC/Makefile | sed 's/\.hip//g')
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=200000 -DRTPOSE_DEV_BUILD -Iinclude -I$SRC"
objs=""
for f in $FILES; do
  o=tools/exp/objs/$f.o
  # the decoder is built without the vectorisers (no packed-fp32 VALU: csrc/Makefile); DECODE_FLAGS="" gives the round-5 form back
  case $f in decode|legacy_pafprocess) X="${DECODE_FLAGS--fno-slp-vectorize -fno-vectorize}";; *) X="";; esac
  if [ -n "$ONLY" ]; then
    if [ "$f" = "$ONLY" ]; then
      o=tools/exp/objs/$f.$(basename $OUT .so).o
      $CC $X "$@" -c $SRC/$f.hip -o $o &
    elif [ ! -f $o ] || [ $SRC/$f.hip -nt $o ]; then
      $CC $X -c $SRC/$f.hip -o $o &
    fi
  else
    o=tools/exp/objs/$f.all.o
    $CC $X "$@" -c $SRC/$f.hip -o $o &
  fi
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs
ls -la $OUT
