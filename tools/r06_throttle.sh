#!/bin/bash
# is the host process CPU-throttled by its cgroup while the inflation leg runs?  (cpu.max = quota period; cpu.stat counts throttled periods)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; do [ -r $f ] && echo "$f: $(cat $f)"; done
nproc; grep -c processor /proc/cpuinfo
stat() { for f in /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat; do [ -r $f ] && grep -h "throttled\|nr_periods" $f | tr '\n' ' '; done; echo; }
for i in 1 2 3 4 5 6 7 8; do
  echo "before: $(stat)"
  MNAV_TRACE=1 timeout 300 python bench.py --skip-c4 --no-cpu > $O/thr_line.json 2> $O/thr_err.txt
  echo "after:  $(stat)"
  python -c "
import json; d=json.loads(open('$O/thr_line.json').read().strip().splitlines()[-1]); c=d['configs']['C3']; print('ms_wave', c['cost_stack_as_specified']['inflation_wave']['ms_wave_of_each_run'], c['cost_stack_used']['inflation_wave']['ms_wave_of_each_run'])"
done
