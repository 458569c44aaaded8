#!/bin/bash
# bench.py twice on one box: default path and with the environment given as arguments (e.g. SD_PF_JOIN=0); prints both lines' essentials
# usage: tools/bench_ab.sh TAG VAR=VALUE...     (from the repo root on the GPU box)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --no-p1000 --no-cpu > $O/bench_a.json 2> $O/bench_a.err
env "$@" python bench.py --no-p1000 --no-cpu > $O/bench_b.json 2> $O/bench_b.err
python - $O "$*" <<'PY'
import json, sys
o = sys.argv[1]
for n, what in (("a", "default"), ("b", sys.argv[2])):
    try:
        d = json.load(open("%s/bench_%s.json" % (o, n)))
    except Exception as e:
        print(n, what, "FAILED", e); print(open("%s/bench_%s.err" % (o, n)).read()[-1500:]); continue
    print(what, "value %.1f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_wall_s"].items()})
    ks = d["kernels"]
    print("   ", {k: round(v["ms"]) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"]) if not k.startswith("host")})
PY
