#!/bin/bash
# round-3 job 1: reduce kernel A/B, the GPU suite, rocprof stats of the default bench with both forms of the reduce
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
python3 tools/reduce_ab.py > gpurun_out/r3_reduce_ab.log 2>&1 || { tail -20 gpurun_out/r3_reduce_ab.log; exit 1; }
cat gpurun_out/r3_reduce_ab.log
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests1.log 2>&1; rc=$?
tail -5 gpurun_out/r3_gpu_tests1.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  rm -rf gpurun_out/prof_r3_v$v
  DMF_REDUCE_V1=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_v$v -- python3 bench.py --no-cpu --steps 1000 --warmup 100 --kappa-steps 0 > gpurun_out/r3_prof_bench_v$v.log 2>&1 || exit 1
  tail -1 gpurun_out/r3_prof_bench_v$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('v1=$v', d['ms_per_step']*1e3, 'us/step', d['value']/1e6, 'M/s kernel', d['roofline']['kernel_ms']*1e3)"
  f=$(find gpurun_out/prof_r3_v$v -name '*kernel_stats.csv' | head -1)
  head -4 "$f" | cut -c1-200
done
