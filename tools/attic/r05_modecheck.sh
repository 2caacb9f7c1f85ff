#!/bin/bash
O=gpurun_out/r05r
mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode train           ', d['value'], d['ms_per_step'])" >> $O/out.txt
timeout 600 python bench.py --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline --no-secondary --sample-images 128 --sample-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode all (short sample)', d['value'], d['ms_per_step'])" >> $O/out.txt
VILLAN_PRESPLIT=0 timeout 600 python bench.py --mode train --no-cpu --no-exact --no-f16 --no-ddp-path --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode train, presplit off', d['value'], d['ms_per_step'])" >> $O/out.txt
done
cat $O/out.txt
