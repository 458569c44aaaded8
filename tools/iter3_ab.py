#!/usr/bin/env python3
"""A/B of the in-memory iterative clustersearch's knobs on ONE set of DBs (1 000 target proteomes, q query proteomes):
usage: python tools/iter3_ab.py q VARIANT [VARIANT ...]     VARIANT = name[:ENV=VALUE[,ENV=VALUE...]]
The first run (tools/iter3_scale.py) builds the DBs, times the default and checks it; every variant is then timed on the same DBs
and its TSV compared with the default's."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def md5(path):
    h = hashlib.md5()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 24), b''):
            h.update(blk)
    return h.hexdigest()


def main():
    import iter3_scale
    from dbutil import SDGPU
    q = int(sys.argv[1])
    work = '/tmp/sd_iter3_ab'
    r = iter3_scale.run(1000, q, 8, keep_dir=work, log=lambda *a: None)
    base = md5(os.path.join(work, 'iter3.tsv'))
    print(json.dumps(dict(variant='default (iter3_scale)', wall_s=round(r['wall_s'], 2), genome_pairs_per_s=round(r['genome_pairs_per_s'], 1),
                          chain_equal=r['module_chain']['tsv_equals_head_of_in_memory_tsv'], parity=r.get('parity_check'), stages=r['stages'][-1])), flush=True)
    for var in sys.argv[2:]:
        name, _, envs = var.partition(':')
        env = dict(os.environ)
        for kv in filter(None, envs.split(',')):
            k, _, v = kv.partition('=')
            env[k] = v
        out = os.path.join(work, 'ab_%s.tsv' % name)
        t0 = time.time()
        p = subprocess.run([SDGPU, 'clustersearch', os.path.join(work, 'Q'), os.path.join(work, 'T'), out, os.path.join(work, 'tmp_' + name),
                            '--num-iterations', '3', '--threads', str(r['threads']), '-v', '3'], capture_output=True, text=True, env=env)
        wall = time.time() - t0
        last = [l for l in p.stdout.splitlines() if l.startswith('in-memory iterations')]
        print(json.dumps(dict(variant=name, env=envs, rc=p.returncode, wall_s=round(wall, 2), genome_pairs_per_s=round(q * 1000 / wall, 1),
                              same_tsv=(md5(out) == base) if p.returncode == 0 else None, stages=last[-1] if last else p.stderr[-300:])), flush=True)


if __name__ == '__main__':
    main()
