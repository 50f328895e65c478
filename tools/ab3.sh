#!/bin/bash
# A/B/C of builds over several bench configurations: tools/ab3.sh "libA.so libB.so ..." -- one line per (lib, args)
LIBS=$1
for args in "--config 1" "--batch 1024 --steps 300" "--config 4 --half 0" "--config 3"; do
  for L in $LIBS; do
    DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L python3 bench.py --no-cpu --kappa-steps 0 --steps 600 --warmup 60 $args 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-22s %-28s us/step %.2f  kernel %.2f us  value %.2f M/s' % ('$L', '$args', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value']/1e6))
"
  done
done
