mkdir -p gpurun_out/wv
for v in ${WV:-0 1 2 3 4 5}; do
  ASB_DOUBLE_PANELS=1 ASB_WIDE_VARIANT=$v python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/wv/w$v.json 2>gpurun_out/wv/w$v.err || exit 1
done
ASB_L2_VARIANT=6 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/wv/s6.json 2>gpurun_out/wv/s6.err
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/wv/s4.json 2>gpurun_out/wv/s4.err
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/wv/*.json")):
    b=json.load(open(f)); print(f.split("/")[-1],round(b["ms_per_step"],3),b["roofline"]["panels_per_step"],round(b["roofline"]["avg_launch_ms"],4))
P
