#!/bin/bash
# bench.py (no CPU leg, no p1000) once per environment setting given as an argument ("-" = default); prints value and stage walls
# usage: tools/bench_env.sh TAG "VAR=V VAR2=V2" "-" ...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for e in "$@"; do
    i=$((i+1))
    if [ "$e" = "-" ]; then python bench.py --no-p1000 --no-cpu > $O/b$i.json 2> $O/b$i.err
    else env $e python bench.py --no-p1000 --no-cpu > $O/b$i.json 2> $O/b$i.err; fi
    python - $O/b$i.json "$e" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "value %.1f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_wall_s"].items() if k in ("prefilter", "align", "prefilter_wait", "total")})
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
done
