#!/bin/bash
# round 5, late A/B at 1 000 proteomes on one box: chunk sizes under the equal-chunk rule (a 3 000-query range: 1 x 3 000, 2 x 1 500 = default,
# 3 x 1 000, 4 x 750) and a third alignment lane; the default interleaved
O=gpurun_out/r05l; mkdir -p $O
run() {   # name, env, bench flags
    env $2 python bench.py --no-children --no-cpu --steps 12 --warmup 3 $3 --detail-out $O/d_$1.json > $O/b_$1.json 2> $O/b_$1.err
    python - $O/b_$1.json "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("%-22s %8.1f genome-pairs/s %8.1f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]), {k: round(v, 1) for k, v in d["stage_wall_s"].items() if k in ("prefilter", "align", "prefilter_wait", "aggregate", "total")})
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run default_a "A=1" ""
run chunk1000 "A=1" "--chunk 1000"
run chunk3000 "A=1" "--chunk 3000"
run align3 "SD_ALIGN_LANES=3" ""
run default_b "A=1" ""
run chunk750 "A=1" "--chunk 750"
run chunk1000_b "A=1" "--chunk 1000"
run default_c "A=1" ""
