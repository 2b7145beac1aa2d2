#!/bin/bash
# configs[4] with different numbers of probe-chain rounds: ms per step, shade / probe stage times (bench.py's exclusive leg)
for R in "$@"; do
  PTR_SSS_CHAINS=$R timeout -k 10 200 python3 bench.py --no-cpu --config 4 --steps 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels_ms_exclusive_per_step']
print('chains $R: %.1f Mrays/s %.1f ms/step  shade %.1f probe %.1f shadow %.1f mis %.1f finish %.1f' % (d['value'], d['ms_per_step'], k['shade'], k.get('sss_probe',0), k['nee_shadow'], k['nee_mis_any'], k['finish']))" || exit 1
done
