#!/bin/bash
# round-3 evidence on the GPU box (run through gpurun from the repo root); outputs under gpurun_out/, tools/summarize_profiles.py r3
# turns them into profiles/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/prof_r3 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_valu4 gpurun_out/pmc_valupan gpurun_out/pmc_valu1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3 -- python3 bench.py --no-cpu --steps 2000 --warmup 200 --kappa-steps 0 > gpurun_out/r3_prof_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu --steps 200 --warmup 20 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r3_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu --steps 200 --warmup 20 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r3_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_valu1 -- python3 bench.py --no-cpu --steps 100 --warmup 10 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r3_pmc_valu1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_valu4 -- python3 bench.py --config 4 --no-cpu --steps 100 --warmup 10 --steps-per-graph 0 > gpurun_out/r3_pmc_valu4.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_valupan -- python3 bench.py --config panms --no-cpu --steps 100 --warmup 10 --steps-per-graph 0 --kappa-steps 0 > gpurun_out/r3_pmc_valupan.log 2>&1 || exit 1
python3 tools/phase_profile_v2.py 256 > gpurun_out/r3_phase_stamps.txt 2>&1 || exit 1
python3 tools/reduce_phase_profile.py > gpurun_out/r3_reduce_stamps.txt 2>&1 || exit 1
echo collected
bash tools/collect_r3_configs.sh > gpurun_out/r3_collect_configs.log 2>&1 || exit 1
rm -rf gpurun_out/prof_r3attn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3attn -- python3 bench.py --config 2 --no-cpu --steps 300 --warmup 30 --kappa-steps 0 > gpurun_out/r3_attn_bench.log 2>&1 || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --kappa-steps 0 2>/dev/null | grep '^{' > gpurun_out/r3_bench_line_steps20.json
echo collected-all
