cd $GRAFT_REPO_ROOT
for i in 1 2; do
for off in 0 4096 2048 6144; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --routes-off $off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['protocol']
print('off=$off steps20', d['ms_per_step'], 'median', p['ms_per_step_median'], 'one_stream', p['ms_per_step_one_stream'], 'kernel', d['roofline']['kernel_ms'], 'k_timed', d['roofline']['kernel_ms_timed_region'])"
done
done
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['protocol']
print('steps100', d['ms_per_step'], 'median', p['ms_per_step_median'])"
