#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/r06; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for L in ${LIBS:-tools/_variants/libmnav_fintm.so}; do
  rm -rf /tmp/fp; MNAV_LIB=$R/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --no-configs > /tmp/fp.log 2>&1
  echo "== $L"; tail -2 /tmp/fp.log | cut -c1-300
  f=$(find /tmp/fp -name '*kernel_stats.csv' | head -1); grep -i "finalize\|tbv_solve\|k_tb_fill" $f | cut -c1-160
done
