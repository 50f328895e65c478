#!/bin/bash
# A/B of two builds of libdmf_hip.so on the same box: tools/ab.sh libA.so libB.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L python3 bench.py --no-cpu --steps 400 --warmup 40 "$@" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-28s ms_per_step %.3f us  kernel(HIP events) %.3f us  value %.2f M/s' % ('$L', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value']/1e6))
"
  done
done
