#!/bin/bash
# A/B/n of builds of libdmf_hip.so on the same box: tools/abn.sh "libA.so libB.so ..." [bench args]
LIBS=$1; shift
for rep in 1 2 3; do
  for L in $LIBS; do
    DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L python3 bench.py --no-cpu --steps 2000 --warmup 200 --kappa-steps 0 "$@" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-28s us/step %.3f  kernel(HIP events) %.3f us  value %.2f M/s' % ('$L', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value']/1e6))
"
  done
done
