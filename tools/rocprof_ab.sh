#!/bin/bash
# rocprofv3 kernel averages of the default bench for several builds: tools/rocprof_ab.sh "libA.so libB.so"
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for L in $1; do
  rm -rf gpurun_out/prof_ab_$L
  DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$L -- python3 bench.py --no-cpu --steps 2000 --warmup 200 --kappa-steps 0 > gpurun_out/prof_ab_$L.log 2>&1 || exit 1
  f=$(ls gpurun_out/prof_ab_$L/*/*_kernel_stats.csv | head -1)
  echo "$L: $(grep 'patch_v2_kernel' $f | head -1 | awk -F, '{print $(NF-6), $(NF-4)}')  reduce: $(grep 'grad_reduce_kernel' $f | head -1 | awk -F, '{print $(NF-4)}')"
done
