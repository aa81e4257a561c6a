#!/bin/bash
# Samples the GPU's clocks / power while the bench loops (is the multi-tile kernel clock-limited?).  Usage: tools/sample_clocks.sh <outdir>
out=${1:-gpurun_out/clk}; mkdir -p $out
python bench.py --steps 2500 --warmup 2 --no-cpu-baseline --no-other-configs > $out/bench.json 2> $out/bench.err &
pid=$!
: > $out/samples.txt
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | tr -s '\t ' ' ' | tr '\n' '|' >> $out/samples.txt
  echo >> $out/samples.txt
  sleep 0.25
done
wait $pid
python - "$out" <<'P'
import re, sys, json
out = sys.argv[1]
rows = []
for ln in open(out + "/samples.txt"):
    s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", ln); m = re.search(r"mclk clock level: \S+ \((\d+)Mhz\)", ln); p = re.search(r"Power \(W\): ([\d.]+)", ln)
    if s and p: rows.append((int(s.group(1)), int(m.group(1)) if m else 0, float(p.group(1))))
busy = [r for r in rows if r[2] > 400]
print("samples", len(rows), "busy", len(busy))
if busy:
    print("busy sclk MHz: min %d max %d | power W: min %.0f max %.0f | mclk %s" % (min(r[0] for r in busy), max(r[0] for r in busy), min(r[2] for r in busy), max(r[2] for r in busy), sorted(set(r[1] for r in busy))))
print("all sclk values:", sorted(set(r[0] for r in rows)))
b = json.load(open(out + "/bench.json")); print("ms/step", b["ms_per_step"], "launch ms", b["roofline"]["avg_launch_ms"])
P
