#!/bin/bash
# bench.py (no CPU leg, no child records) once per environment setting given as an argument ("-" = default); prints value, stage walls,
# the in-pipeline and the isolated prefilter kernel times.  BENCH_ARGS adds bench flags (e.g. "--proteomes 100 --steps 10").
# usage: tools/bench_env.sh TAG "VAR=V VAR2=V2" "-" ...      (from the repo root on the GPU box)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
i=0
for e in "$@"; do
    i=$((i+1))
    if [ "$e" = "-" ]; then python bench.py --no-children --no-cpu ${BENCH_ARGS:-} --detail-out $O/d$i.json > $O/b$i.json 2> $O/b$i.err
    else env $e python bench.py --no-children --no-cpu ${BENCH_ARGS:-} --detail-out $O/d$i.json > $O/b$i.json 2> $O/b$i.err; fi
    python - $O/b$i.json $O/d$i.json "$e" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    m = json.load(open(sys.argv[2]))['main']
    print(sys.argv[3], "value %.1f" % d["value"], "ms/step %.1f" % d["ms_per_step"],
          {k: round(v, 2) for k, v in d["stage_wall_s"].items() if k in ("prefilter", "align", "prefilter_wait", "total")},
          {k: round(v) for k, v in d["roofline"]["stage_kernel_ms"].items()})
    ks = m["kernels"]
    print("    pipeline ms:", {k.replace('prefilter_', ''): round(v["ms"]) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"]) if k.startswith("prefilter_")})
    iso = m["roofline"].get("isolated") or {}
    print("    isolated %s queries: %.1f ms" % (iso.get("queries"), iso.get("kernel_ms") or 0.0),
          {k.replace('prefilter_', ''): v for k, v in sorted((iso.get("kernel_ms_by_name") or {}).items(), key=lambda kv: -kv[1])})
except Exception as ex:
    print(sys.argv[3], "FAILED", ex)
    print(open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
done
