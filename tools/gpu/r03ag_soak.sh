#!/bin/bash
# round 3: twelve complete 300-iteration jobs in twelve processes at HEAD: models_md5 of every bench line
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03ag; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 600 python bench.py --no-cpu-baseline --roofline-steps 1 > $O/soak_$i.log 2>&1
  grep '^{"metric' $O/soak_$i.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("job", d["models_md5"], "%.10f" % d["repair_accuracy_vs_clean"], "%.0f cells/s" % d["value"])'
done | tee $O/soak.log
sort $O/soak.log | cut -d" " -f2 | uniq -c
