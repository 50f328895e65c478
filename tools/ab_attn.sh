#!/bin/bash
# A/B of builds on the attention train step: tools/ab_attn.sh "libA.so libB.so ..."
for L in $1; do
  DMF_LIB=$PWD/dual-modal-fusion_amd/dmf/$L python3 bench.py --no-cpu --kappa-steps 0 --steps 200 --warmup 30 --config 2 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-22s us/step %.2f  kernel %.2f us  value %.3f M/s' % ('$L', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['value']/1e6))
"
done
