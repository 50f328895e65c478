#!/bin/bash
# rocprofv3 kernel-trace of bench.py for each given build of the library; prints the patch / reduce kernels' average durations
export TMPDIR=/tmp
for L in "$@"; do
  D=gpurun_out/prof_$L
  rm -rf $D; mkdir -p $D
  DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 bench.py --no-cpu --steps 400 --warmup 40 > $D/bench.log 2>&1
  F=$(find $D -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  if [ -n "$F" ]; then head -4 "$F" | cut -c1-200; else echo "no stats file"; ls -R $D | head; fi
done
