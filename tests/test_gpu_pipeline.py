"""End-to-end on the GPU: the reference's own regression input (examples/*.faa, run_regression.sh) through
prefilter -> align -> aggregation -> clusterhits -> TSV must reproduce the reference's known answers."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from spacedust_amd.pipeline import SetDB, ClusterSearch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_examples(host):
    names, seqs, set_id, pos, strand = [], [], [], [], []
    sources = ['NC_000913.faa', 'NC_000915.faa']
    for si, f in enumerate(sources):
        with gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rt') as fh:
            cur, hdr, idx = [], None, 0
            def flush():
                nonlocal idx
                if hdr is None:
                    return
                w = hdr.replace(' ', '').split('#')          # createsetdb.sh:119-139
                acc, st, en, sd = w[0], int(w[1]), int(w[2]), w[3]
                if sd == '-1':
                    st, en = en, st
                names.append('%s_%d_%d_%d' % (acc, idx, st, en))
                seqs.append(''.join(cur))
                set_id.append(si); pos.append(idx); strand.append(1 if st < en else 0)
                idx += 1
            for line in fh:
                line = line.rstrip('\n')
                if line.startswith('>'):
                    flush()
                    hdr, cur = line[1:], []
                else:
                    cur.append(line)
            flush()
    res, off = host.map_sequences(seqs)
    return SetDB(res, off, set_id, pos, strand, 2, names=names, sources=sources)


def test_examples_regression_known_answers(gpu, host, tmp_path):
    db = load_examples(host)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2)
    out = cs.search(db, same_db=True, tsv_path=str(tmp_path / 'result.tsv'), canonical=True, chunk_queries=3000)
    assert cs.index.n_entries == 1784989 and cs.index.masked_residues == 11546
    assert cs.stats['prefilter_hits'] == 98957
    assert out['accepted'] == 15065
    lines = open(tmp_path / 'result.tsv').readlines()
    n_clu = sum(1 for l in lines if l.count('\t') == 4)
    n_hit = len(lines) - n_clu
    sig = sum(1 for l in lines if l.count('\t') == 4 and float(l.split('\t')[2]) < 1e-20)
    # R/util/run_regression.sh:20-23: 308 hit lines, 2 clusters with P < 1E-20 (108 clusters in total)
    assert (n_hit, sig, n_clu) == (308, 2, 108)
    lines.sort(key=lambda s: s.encode())
    assert hashlib.md5(''.join(lines).encode()).hexdigest() == 'abb28ee37bc130a5f09a9f767ef00ccf'
