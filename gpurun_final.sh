set -x
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --workload hydro_plm_hllc_rk2_256 --no-cpu-baseline > gpurun_out/bench_hydro.json 2>/dev/null
python bench.py --workload mhd_wenoz_hlld_rk3_256 --no-cpu-baseline > gpurun_out/bench_wenoz.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
tail -3 gpurun_out/gpu_tests.log
