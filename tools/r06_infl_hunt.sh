#!/bin/bash
# the inflation wave's rare slow mode (72-85 ms instead of 5): run the bench's legs up to C3 again and again with the library's
# trace, which carries host-side timings of every chunk (launch / wait); keep the first slow instance
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06/infl_hunt; rm -rf $O; mkdir -p $O
for i in $(seq 1 ${HUNT_N:-18}); do
  SPIN=0; [ $i -gt ${HUNT_SPIN_AFTER:-999} ] && SPIN=1                  # (the second half with the host polling instead of blocking: MNAV_SPIN_WAIT)
  MNAV_SPIN_WAIT=$SPIN MNAV_TRACE=1 timeout 300 python bench.py --skip-c4 --no-cpu > $O/line.json 2> $O/err.txt
  python - <<PY
import json
d = json.loads(open("$O/line.json").read().strip().splitlines()[-1])
c = d["configs"]["C3"]
a, b = c["cost_stack_as_specified"]["inflation_wave"], c["cost_stack_used"]["inflation_wave"]
print("run $i spin $SPIN ms_wave of each run", a["ms_wave_of_each_run"], b["ms_wave_of_each_run"])
PY
  grep -h "inflation it" $O/err.txt | sed 's/.*this chunk/chunk/' | awk '{ if ($0 ~ /wait [2-9][0-9]\./ || $0 ~ /wait [0-9][0-9][0-9]/) print "   SLOW CHUNK: " $0 }'
  slow=$(python -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); c=d['configs']['C3']; print(int(max(c['cost_stack_as_specified']['inflation_wave']['ms_wave_of_each_run'] + c['cost_stack_used']['inflation_wave']['ms_wave_of_each_run']) > 40))")
  if [ "$slow" = 1 ]; then cp $O/err.txt $O/slow_err_$i.txt; cp $O/line.json $O/slow_line_$i.json; echo "SLOW at run $i"; fi
done
