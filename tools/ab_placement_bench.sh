# the headline line of bench.py with the three placements of its resident arrays, interleaved in ONE call
set -e
mkdir -p gpurun_out/r4c
for i in 1 2 3; do
  for pl in torch plain tuned; do
    python bench.py --no-cpu-baseline --no-secondary --steps 100 --placement $pl "$@" 2>gpurun_out/r4c/err_${pl}_$i.log | tail -1 > gpurun_out/r4c/${pl}_$i.json
  done
done
python - <<'PY'
import json
for pl in ("torch","plain","tuned"):
    r=[json.load(open("gpurun_out/r4c/%s_%d.json"%(pl,i))) for i in (1,2,3)]
    print("%-6s"%pl," ".join("%.4f (min %.4f, frac %.3f)"%(d["ms_per_step"],d["roofline"]["kernel_ms_min"],d["roofline"]["frac"]) for d in r), " floors", " ".join("%.3f"%d["roofline"]["mixed_read_write_floor"]["ms"] for d in r))
print(json.load(open("gpurun_out/r4c/tuned_1.json"))["config"]["placement"])
PY
