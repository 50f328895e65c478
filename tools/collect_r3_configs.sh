#!/bin/bash
# one bench line per BASELINE configuration / batch size (the loop of tools/r3_job.sh alone); tools/summarize_profiles.py's
# configs_md() turns gpurun_out/r3_config_lines.jsonl into profiles/r3_configs.md
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
: > gpurun_out/r3_config_lines.jsonl
for a in "--config 2 --steps 300 --warmup 30" "--config 3 --steps 1000 --warmup 100" "--config 4 --steps 1000 --warmup 100" "--config 4 --half 0 --steps 1000 --warmup 100" "--config panms --steps 1000 --warmup 100" "--half 1 --steps 1000 --warmup 100" "--batch 512 --steps 500" "--batch 1024 --steps 500" "--batch 4096 --steps 200"; do
  echo "# bench.py $a --no-cpu --kappa-steps 0" >> gpurun_out/r3_config_lines.jsonl
  timeout -k 10 300 python3 bench.py $a --no-cpu --kappa-steps 0 2>gpurun_out/r3_cfg_err.log | tail -1 >> gpurun_out/r3_config_lines.jsonl || exit 1
  echo "done: $a"
done
echo collected
