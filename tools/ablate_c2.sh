for ab in 0 2 4 6; do
  PFMSCAN_ABLATE=$ab python bench.py --workload c2 --width 8 --no-cpu-baseline --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate $ab: %.4f ms (min %.4f)' % (d['ms_per_step'], d['roofline']['kernel_ms_min']))"
done
