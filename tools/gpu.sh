#!/bin/bash
# gpurun with patience: repeats a call that came back "transient" (no slot / no box: nothing charged), never one that ran.
#   tools/gpu.sh LOGFILE TIMEOUT 'command'
log=$1; to=$2; shift 2
make -C "$(dirname "$0")/../animsnapbases_amd/csrc" -j8 2>&1 | grep -E "error|warning" -A4 | head -20      # never send a stale library
for try in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  if grep -q "status=transient" $log; then sleep 90; continue; fi
  break
done
