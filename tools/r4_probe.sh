#!/bin/bash
# round 4: the structured-data kinds of config 4's shape -- time and reads per kind (plain run), then the per-read diagnostics
# (ASB_DEBUG_PANELS=1; not a timing), the residual loop beside the rank-deficient kind.  Optional pytest selection as further args.
out=gpurun_out/${1:-r4a}; shift; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
if [ $# -gt 0 ]; then
  timeout -k 10 900 python -m pytest "$@" -q -m gpu -x > $out/tests.log 2>&1; rc=$?
  tail -n 15 $out/tests.log; ok $rc || exit 1
fi
timeout -k 10 400 python tools/structured_probe.py ${KINDS:-lowrank slow bumps rankdef} > $out/probe.log 2> $out/probe.err; rc=$?
cat $out/probe.log; ok $rc || exit 1
PROBE_MODE=residual timeout -k 10 300 python tools/structured_probe.py rankdef > $out/probe_residual.log 2> $out/probe_residual.err; rc=$?
cat $out/probe_residual.log; ok $rc || exit 1
ASB_DEBUG_PANELS=1 timeout -k 10 400 python tools/structured_probe.py ${KINDS:-lowrank slow bumps rankdef} > $out/probe_dbg.log 2> $out/probe_dbg.err; rc=$?
grep -c "read at" $out/probe_dbg.err
