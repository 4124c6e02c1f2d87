cd $GRAFT_REPO_ROOT
B="python bench.py --no-other-configs --no-cpu-baseline --experiment"
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'value %.3f G' % (d['value'] / 1e9), 'ms/step', d['ms_per_step'], 'scan ms', d['roofline'].get('avg_kernel_ms'), 'serial', d.get('serial_ms_per_sample'))
"; }
$B 2>/dev/null | show base
for n in 192 208 176 224; do BK_CU_SPLIT=$n $B 2>/dev/null | show split$n; done
BK_CU_SPLIT=192 BK_CU_SPLIT_SCAN_ONLY=1 $B 2>/dev/null | show scanonly192
$B 2>/dev/null | show base
