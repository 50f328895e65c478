#!/bin/bash
# round-2 evidence on the GPU box (run through gpurun from the repo root): rocprofv3 stats, the two PMC passes, phase stamps,
# one bench line per BASELINE configuration.  Outputs under gpurun_out/; tools/summarize_profiles.py r2 turns them into profiles/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/prof_r2 gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -- python3 bench.py --no-cpu --steps 2000 --warmup 200 --kappa-steps 0 > gpurun_out/r2_prof_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu --steps 200 --warmup 20 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r2_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu --steps 200 --warmup 20 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r2_pmc_write.log 2>&1 || exit 1
python3 tools/phase_profile_v2.py 256 > gpurun_out/r2_phase_stamps.txt 2>&1 || exit 1
: > gpurun_out/r2_config_lines.jsonl
for a in "--config 2 --steps 300 --warmup 30" "--config 3 --steps 1000 --warmup 100" "--config 4 --steps 1000 --warmup 100" "--config 4 --half 0 --steps 1000 --warmup 100" "--config panms --steps 1000 --warmup 100" "--half 1 --steps 1000 --warmup 100" "--batch 512 --steps 500" "--batch 1024 --steps 500" "--batch 4096 --steps 200"; do
  echo "# bench.py $a --no-cpu --kappa-steps 0" >> gpurun_out/r2_config_lines.jsonl
  timeout -k 10 300 python3 bench.py $a --no-cpu --kappa-steps 0 2>gpurun_out/r2_cfg_err.log | tail -1 >> gpurun_out/r2_config_lines.jsonl || exit 1
done
echo collected
