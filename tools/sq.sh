#!/bin/bash
# usage: tools/sq.sh <tag> [kernel substrings...]   -- SQ stall counters of the bench command
tag=$1; shift
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/prof_$tag -o s -- python bench.py --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/prof_$tag/s_results.db "$@" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['kernel'][10:52], r['calls'], '%.0f us' % r['avg_us'], {k[3:]: '%.3g' % v for k, v in r.items() if k.startswith('SQ_') and not k.endswith('_per_ns')})
"
