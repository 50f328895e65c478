#!/bin/bash
# rebuild libdmf_hip.so and the stamps variant from the repo root, whatever the caller's cwd is
cd "$(dirname "$0")/.." && python dual-modal-fusion_amd/build.py --force --stamps 2>&1 | grep -v "^/opt/rocm/bin/hipcc"
ls -la --time-style=+%T dual-modal-fusion_amd/dmf/libdmf_hip.so | awk '{print "built", $6}'
