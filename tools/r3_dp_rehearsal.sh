#!/bin/bash
# 2-rank rehearsals of bench.py on a ONE-GPU box, started WITHOUT a launcher (bench.py spawns its own ranks): all ranks on
# cuda:0, gloo as the process group's transport.  Small shapes: a spinning reduce kernel and a peer's 150-KB-LDS patch kernel of
# the headline shape cannot be co-resident on one GPU.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
export DMF_SINGLE_DEVICE=1 DMF_DIST_BACKEND=gloo DMF_XGMI_TIMEOUT_MS=4000
echo "== config 1 shape family, 2 ranks, one-shot exchange (admitted per run) or the group's all_reduce"
timeout -k 10 300 python3 bench.py --gpus 2 --size 40 --bands 8 --patch 5 --batch 64 --classes 4 --steps 40 --warmup 8 --steps-per-graph 10 --kappa-steps 0 --no-cpu 2>gpurun_out/r3_dp2_err.log | tail -1
tail -3 gpurun_out/r3_dp2_err.log | cut -c1-300
echo "== the same with DMF_ALLREDUCE=rccl (here: gloo) — the fallback path"
DMF_ALLREDUCE=rccl timeout -k 10 300 python3 bench.py --gpus 2 --size 40 --bands 8 --patch 5 --batch 64 --classes 4 --steps 40 --warmup 8 --kappa-steps 0 --no-cpu 2>gpurun_out/r3_dp2b_err.log | tail -1
echo "== config 4 (stage 2), 2 ranks"
timeout -k 10 300 python3 bench.py --gpus 2 --config 4 --size 40 --patch 5 --batch 32 --classes 4 --steps 20 --warmup 4 --no-cpu --half 0 2>gpurun_out/r3_dp2c_err.log | tail -1
tail -3 gpurun_out/r3_dp2c_err.log | cut -c1-300
echo "== the FULL-size configuration (batch 256 per rank, 11x11x200), 2 and 4 ranks, every rank on its own share of the compute units (DMF_CU_SHARE=1: dmf.xgmi.cu_share_stream) — functional, not a timing"
for n in 2 4; do
  DMF_CU_SHARE=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 200 --warmup 20 --kappa-steps 0 --no-cpu 2>gpurun_out/r3_dpfull${n}_err.log | tail -1
done
