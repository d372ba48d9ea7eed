set -e
mkdir -p gpurun_out/ab_hits
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_library.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  PFMSCAN_PROFILE_GENERIC=1 python bench.py --mode hits --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > gpurun_out/ab_hits/g_$i.json
  python bench.py --mode hits --no-cpu-baseline --steps 100 2>/dev/null | tail -1 > gpurun_out/ab_hits/f_$i.json
done
python - <<'PY'
import json
for k in ("g","f"):
    r=[json.load(open("gpurun_out/ab_hits/%s_%d.json"%(k,i))) for i in (1,2,3)]
    print(k," ".join("%.4f (min %.4f) hits %s"%(d["ms_per_step"],d["roofline"]["kernel_ms_min"],d["config"]["hits_per_step"]) for d in r))
PY
