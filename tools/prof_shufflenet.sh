#!/bin/bash
# rocprofv3 evidence for BASELINE configs[3] (ShuffleNetV2 x1.0, 128 x 368 x 368): kernel trace + stats,
# then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md).  usage: prof_shufflenet.sh TAG [dtype]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
TAG=${1:-r02}; DT=${2:-fp32}
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/${TAG}_sn_trace -o t -- python $R/tools/bench_shufflenet.py 128 5 $DT > $O/${TAG}_sn_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/${TAG}_sn_$c -o t -- python $R/tools/bench_shufflenet.py 128 2 $DT > $O/${TAG}_sn_$c.log 2>&1
done
for d in ${TAG}_sn_trace ${TAG}_sn_FETCH_SIZE ${TAG}_sn_WRITE_SIZE; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/$d.summary.txt 2>&1
  rm -rf $O/$d
done
