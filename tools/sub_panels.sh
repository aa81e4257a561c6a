# A/B of the sub-panels per read of X: "<ASB_SUB_PANELS> <ASB_SUB_FIRST> <ASB_WIDE_VARIANT>" per run
mkdir -p gpurun_out/sp
ASB_WIDE_VARIANT=${TV:-4} ASB_SUB_PANELS=${TS:-8} ASB_SUB_FIRST=${TF:-3} timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "double or timeout or guess or spec" > gpurun_out/sp/t.log 2>&1; rc=$?; tail -2 gpurun_out/sp/t.log
[ $rc -eq 124 ] && exit 1
IFS=';'
for cfg in ${SPC:-"3 3 4;8 3 4;4 4 4"}; do
  IFS=' '; set -- $cfg
  ASB_SUB_PANELS=$1 ASB_SUB_FIRST=$2 ASB_WIDE_VARIANT=$3 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > gpurun_out/sp/s$1_f$2_w$3.json 2>gpurun_out/sp/s$1_f$2_w$3.err || { tail -3 gpurun_out/sp/s$1_f$2_w$3.err; exit 1; }
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/sp/s*_f*.json")):
    b=json.load(open(f)); print(f.split("/")[-1],round(b["ms_per_step"],3),b["roofline"]["panels_per_step"],round(b["roofline"]["avg_launch_ms"],4))
P
