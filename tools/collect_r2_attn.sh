#!/bin/bash
# attention evidence on the GPU box: rocprofv3 stats of bench.py --config 2, the SQ counters of attn_train_kernel (own passes),
# one bench line.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/prof_c2 gpurun_out/pmc_c2a gpurun_out/pmc_c2b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c2 -- python3 bench.py --config 2 --no-cpu --steps 200 --warmup 20 --kappa-steps 0 > gpurun_out/c2_prof.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_c2a -- python3 bench.py --config 2 --steps 40 --warmup 10 --steps-per-graph 0 --no-cpu --kappa-steps 0 > gpurun_out/c2_pmca.log 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_c2b -- python3 bench.py --config 2 --steps 40 --warmup 10 --steps-per-graph 0 --no-cpu --kappa-steps 0 > gpurun_out/c2_pmcb.log 2>&1 || exit 1
python3 bench.py --config 2 --no-cpu --kappa-steps 0 --steps 300 --warmup 30 2>/dev/null | tail -1 > gpurun_out/c2_line.json
echo collected
