#!/bin/bash
# The two PMC passes behind `roofline.traffic` (separate --pmc runs, kernel-trace off: gpurun refuses --pmc with other trace domains) + tools/update_traffic.py:
#   tools/traffic_pass.sh <tag>     -> gpurun_out/<tag>/traffic.json (copy to profiles/traffic.json: it is stamped with this build's kernel-source sha)
TAG=${1:-tp}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
cd $R
python tools/update_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $TAG > $O/traffic_update.txt 2>&1
cp profiles/traffic.json $O/traffic.json; rm -rf $O/fetch $O/write; cat $O/traffic_update.txt | head -8
