#!/bin/bash
# Round 5, GPU call 4: the whole GPU suite (all failures listed), then the round-5 profile collection
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "^\.\|^$" | tail -60 ) > gpurun_out/r05_pytest4.txt 2>&1
grep "passed\|failed" gpurun_out/r05_pytest4.txt | tail -2
bash tools/profile_r05.sh > gpurun_out/r05_profile.log 2>&1
tail -5 gpurun_out/r05_profile.log
