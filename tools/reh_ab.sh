#!/bin/bash
# Development aid (GPU box, repo root): N = 1 / rehearsal with and without the x1 strips in the buffers, interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  REHEARSE=0 python tools/reh_run.py
  APK_X1_DIRECT=0 python tools/reh_run.py
  APK_X1_DIRECT=1 python tools/reh_run.py
  APK_X1_DIRECT=0 OVERLAP=0 python tools/reh_run.py
  APK_X1_DIRECT=1 OVERLAP=0 python tools/reh_run.py
done 2>&1 | grep -v amdgpu.ids
