# A/B of the projection kernels after a change: single-panel (k_project_l2s, variants 4 / 6), 2 / 3 / 4 sub-panels per read
mkdir -p gpurun_out/ab
timeout -k 10 700 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "double or timeout or guess or spec or project" > gpurun_out/ab/t.log 2>&1; rc=$?; tail -2 gpurun_out/ab/t.log
[ $rc -eq 124 ] && exit 1
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err || { tail -3 gpurun_out/ab/$name.err; exit 1; }; }
run single_v4 ASB_DOUBLE_PANELS=0
run single_v6 ASB_DOUBLE_PANELS=0 ASB_L2_VARIANT=6
run sub2 ASB_SUB_PANELS=2 ASB_SUB_FIRST=2
run sub3 ASB_SUB_PANELS=3 ASB_SUB_FIRST=3
run sub4 ASB_SUB_PANELS=4 ASB_SUB_FIRST=4
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    b=json.load(open(f)); print(f.split("/")[-1],round(b["ms_per_step"],3),b["roofline"]["panels_per_step"],round(b["roofline"]["avg_launch_ms"],4))
P
