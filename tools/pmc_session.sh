#!/bin/bash
# PMC passes for the roofline's `traffic` (guide, HBM section: FETCH_SIZE and WRITE_SIZE in SEPARATE runs, --kernel-trace
# only).  Usage: tools/pmc_session.sh <tag>   -> gpurun_out/<tag>/pmc_fetch, pmc_write (CSV)
tag=${1:-pmc}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$(echo $c | tr A-Z a-z | sed s/_size//)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $out/$c.json 2> $out/$c.err
  rc=$?; echo "$c rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit 1
done
find $out -name "*counter_collection.csv" | head
