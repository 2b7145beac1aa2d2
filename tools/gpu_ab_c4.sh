#!/bin/bash
# A/B of library variants on configs[4] (or "$CFG"): usage tools/gpu_ab_c4.sh <variant|-> ...   ("-" = the default build)
CFG=${CFG:-4}
for V in "$@"; do
  if [ "$V" = "-" ]; then unset PTR_HIP_LIB; else export PTR_HIP_LIB=$GRAFT_REPO_ROOT/pathtracer_amd/lib/libptr_hip_$V.so; fi
  timeout -k 10 200 python3 bench.py --no-cpu --config $CFG --steps 6 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels_ms_exclusive_per_step']
print('variant $V: %.1f Mrays/s %.1f ms/step ' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f' % (a, b) for a, b in k.items()))" || exit 1
done
