for flags in "" "--sustained 0" "--no-copies-base" "--sustained 0 --no-copies-base"; do
python bench.py --no-other-workloads --no-cpu-baseline $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rehearsal_8gpu_rank']
print('$flags', 'headline %.3f n1 %.3f overlapped %.3f sync %.3f general %.3f' % (d['ms_per_step'], r['n1']['ms_per_step'], r['overlapped']['ms_per_step'], r['synchronous']['ms_per_step'], d['roofline']['general_stage']['ms_per_stage']))"
done
