# A/B of the fixed-width k_profile against the width-generic one on C3 at several widths, arrays placed by pfmscan_place_alloc,
# interleaved in ONE call (same box)
set -e
mkdir -p gpurun_out/r4b
for w in ${WIDTHS:-6 8 9 10 11 12 16 18}; do
 for i in 1 2 3; do
  PFMSCAN_PROFILE_GENERIC=1 python bench.py --no-cpu-baseline --no-secondary --steps 100 --width $w 2>/dev/null | tail -1 > gpurun_out/r4b/generic_w${w}_$i.json
  PFMSCAN_PROFILE_FIXED_MIN=0 python bench.py --no-cpu-baseline --no-secondary --steps 100 --width $w 2>/dev/null | tail -1 > gpurun_out/r4b/fixed_w${w}_$i.json
 done
done
python - <<'PY'
import json, os
for w in [int(x) for x in os.environ.get("WIDTHS", "6 8 9 10 11 12 16 18").split()]:
    for k in ("generic","fixed"):
        r=[json.load(open("gpurun_out/r4b/%s_w%d_%d.json"%(k,w,i))) for i in (1,2,3)]
        print("w=%2d %-8s"%(w,k)," ".join("%.4f (min %.4f, frac %.3f)"%(d["ms_per_step"],d["roofline"]["kernel_ms_min"],d["roofline"]["frac"]) for d in r))
PY
