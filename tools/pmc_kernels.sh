#!/bin/bash
# Per-kernel PMC counters of one command, one counter set per run (rocprofv3 --pmc with --kernel-trace only).
# usage: tools/pmc_kernels.sh OUTDIR "COUNTERS1" ["COUNTERS2" ...] -- command...      (run from the repo root on the GPU box)
set -u
OUT=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
R=$(pwd)
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for c in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- "$@" > $OUT/run_$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        m = re.match(r'(?:void )?([A-Za-z0-9_:<>, ]+?)\(', k)
        k = (m.group(1) if m else k)[:48]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
    for k, d in seen:
        pass
    per = defaultdict(set)
    for k, d in seen:
        per[k].add(d)
    for k in per:
        calls[k] = max(calls[k], len(per[k]))
names = sorted({c for d in acc.values() for c in d})
print('%-48s %6s ' % ('kernel', 'calls') + ' '.join('%16s' % n[:16] for n in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print('%-48s %6d ' % (k, calls[k]) + ' '.join('%16.4g' % acc[k].get(n, 0.0) for n in names))
PY
rm -rf $OUT/pmc_*
