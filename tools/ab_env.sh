# A/B of one environment switch on the headline line, interleaved in ONE call:  tools/ab_env.sh VAR=value [bench args]
set -e
kv=$1; shift
mkdir -p gpurun_out/ab_env
for i in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" 2>/dev/null | tail -1 > gpurun_out/ab_env/base_$i.json
  env $kv python bench.py --no-cpu-baseline --no-secondary --steps 100 "$@" 2>/dev/null | tail -1 > gpurun_out/ab_env/alt_$i.json
done
python - "$kv" <<'PY'
import json, sys
for k, name in (("base", "default"), ("alt", sys.argv[1])):
    r=[json.load(open("gpurun_out/ab_env/%s_%d.json"%(k,i))) for i in (1,2,3,4)]
    print("%-22s"%name," ".join("%.4f (min %.4f)"%(d["ms_per_step"],d["roofline"]["kernel_ms_min"]) for d in r))
PY
