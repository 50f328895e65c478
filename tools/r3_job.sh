#!/bin/bash
# round-3 GPU job: the GPU suite, then rocprofv3 stats of the default bench, the reduce and patch stamps (stamps build)
# usage: tools/r3_job.sh TAG [skip-tests]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-x}
if [ -z "$2" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_tests_$TAG.log 2>&1; rc=$?
  tail -15 gpurun_out/r3_gpu_tests_$TAG.log
  [ $rc -eq 0 ] || exit $rc
fi
rm -rf gpurun_out/prof_r3_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_$TAG -- python3 bench.py --no-cpu --steps 1000 --warmup 100 --kappa-steps 0 > gpurun_out/r3_prof_bench_$TAG.log 2>&1 || { tail -5 gpurun_out/r3_prof_bench_$TAG.log; exit 1; }
f=$(find gpurun_out/prof_r3_$TAG -name '*kernel_stats.csv' | head -1)
head -4 "$f" | cut -c1-220
grep '^{' gpurun_out/r3_prof_bench_$TAG.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG (under rocprof)', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,2), 'M/s; kernel (events)', round(d['roofline']['kernel_ms']*1e3,2))"
python3 bench.py --no-cpu --steps 2000 --warmup 200 --kappa-steps 0 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', round(d['ms_per_step']*1e3,2), 'us/step', round(d['value']/1e6,2), 'M/s; kernel (events)', round(d['roofline']['kernel_ms']*1e3,2), 'frac', round(d['roofline']['frac'],3))"
python3 tools/reduce_phase_profile.py > gpurun_out/r3_reduce_stamps_$TAG.txt 2>&1
python3 tools/phase_profile_v2.py 256 > gpurun_out/r3_phase_stamps_$TAG.txt 2>&1
tail -4 gpurun_out/r3_phase_stamps_$TAG.txt
