#!/bin/bash
# power / clock while the north-star stage runs in a loop (run on the GPU box): is the chip at its power cap?
cd $GRAFT_REPO_ROOT
O=gpurun_out/power; mkdir -p $O
rocm-smi --showmaxpower --showpower --showclocks > $O/idle.txt 2>&1
python tools/stage_time.py --gam0 0.5 --fill 2 --dt --reps 1500 > $O/stage.txt 2>&1 &
PID=$!
sleep 3
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" >> $O/busy.txt; echo "--" >> $O/busy.txt; sleep 0.5; done
wait $PID
tools/ubench/ubench_fp64 3000 > /dev/null 2>&1 &
PID=$!
sleep 1.5
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $O/busy_fma.txt; echo "--" >> $O/busy_fma.txt; sleep 0.4; done
wait $PID
cat $O/idle.txt | grep -E "Power|sclk|Max" ; echo ==== ; cat $O/busy.txt | head -40; echo ====; cat $O/busy_fma.txt | head -12; tail -2 $O/stage.txt
