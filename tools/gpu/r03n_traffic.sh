#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
ALL=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$O/pmc_fetch -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_fetch.log 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$O/pmc_write -- python $OLDPWD/tools/probe.py --iters 1 --targets $ALL --stats 0 > $OLDPWD/$O/pmc_write.log 2>&1 )
python tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write r03n > $O/traffic_json.log 2>&1; tail -40 $O/traffic_json.log; cp profiles/traffic.json $O/traffic.json; cp profiles/r03n_hbm_traffic_pmc.txt $O/ 2>/dev/null
grep "^target" $O/pmc_fetch.log | tail -4
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
