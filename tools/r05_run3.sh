#!/bin/bash
# Round 5, GPU call 3: the new C-ABI parity test with tracebacks, the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "takes_its_input_from_the_conserved_state" --tb=short 2>&1 | grep -v "^tests/\|^$" | grep "Error\|assert\|^E \|passed\|failed\|FAILED" | sort | uniq -c | sort -rn | head -40 > gpurun_out/r05_pytest3a.txt 2>&1
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r05_pytest3.txt 2>&1
( time python bench.py --steps 20 ) > gpurun_out/r05_bench3.json 2> gpurun_out/r05_bench3.err
cat gpurun_out/r05_pytest3a.txt; tail -8 gpurun_out/r05_pytest3.txt; tail -4 gpurun_out/r05_bench3.err
