#!/usr/bin/env python3
"""Golden fixtures for the DB boundary of config 1 (the reference's regression input, examples/*.faa, all-vs-all):
  tests/golden/config1_pref.tsv.gz   flattened prefilter DB  (qKey tKey score diag), 98 957 lines
  tests/golden/config1_aln.tsv.gz    flattened alignment DB  (qKey + Matcher::resultToBuffer line), 15 065 lines
in DB order: entries by query key, the lines of an entry in the order the module writes them.
Both are produced by the REAL reference classes (oracle/_ref/libsdref.so: QueryMatcher, SmithWaterman, ALP) driven here, and
both must hash -- after `LC_ALL=C sort` -- to what the reference binary's own DBs hash to:
  pref_0 -> 8109a70bdea70ee10e0dbd27ba6b7e37      result -> 2e917f0e9782e8a7412c7360aa7bf1b4
The script asserts both, so the fixtures are pinned to the binary, not to this repository's formatting.
(The text formatting restates Matcher::resultToBuffer, Matcher.cpp:280-327, including fastSeqIdToBuffer's "1.00" for an
identity of one -- Util.cpp:222-251 returns the position of the terminator in that branch.)"""
import gzip, hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, RefSW, read_fasta
GOLD = os.path.join(ROOT, 'tests', 'golden')
orc = Oracle(8)
ref = Ref(6)
n1, s1 = read_fasta('/root/reference/examples/NC_000913.faa')
n2, s2 = read_fasta('/root/reference/examples/NC_000915.faa')
seqs = s1 + s2
lens = np.array([len(s) for s in seqs])
off = np.zeros(len(seqs) + 1, np.uint64)
off[1:] = np.cumsum(lens)
blob = ''.join(seqs).encode()
dbres = int(lens.sum())
t0 = time.time()
rix = ref.index(blob, off)
rpf = rix.prefilter(int(lens.max()))
pref, pref_lines = {}, []
for q in range(len(seqs)):
    ids, sc, dg = rpf.query(seqs[q], q)[:3]
    ql = np.float32(lens[q])
    for t, s, d in zip(ids, sc, dg):
        if not (np.float32(lens[t]) / ql >= np.float32(0.8)):      # Prefiltering.cpp:856-863 (--cov-mode 2 -c 0.8)
            continue
        pref.setdefault(q, []).append(int(t))
        pref_lines.append('%d\t%d\t%d\t%d\n' % (q, t, s, np.int16(np.uint16(d))))
print('prefilter rows', len(pref_lines), round(time.time() - t0, 1), 's', flush=True)


def seqid_text(seqid):
    if seqid == np.float32(1.0):
        return '1.00'
    return '0.' + ('0' if seqid < np.float32(0.10) else '') + ('0' if seqid < np.float32(0.01) else '') + \
        str(int(np.float32(seqid * np.float32(1000))))


def line(q, t, r, qL, tL):
    """Matcher::getSWResult (Matcher.cpp:88-137) + Alignment::checkCriteria (-e 10 -c 0.8 --cov-mode 2 --min-aln-len 30)"""
    if q == t:
        seqid = np.float32(1.0)
    else:
        if r['tStart'] < 0 or r['qStart'] < 0 or r['btLen'] == 0:
            return None
        alnlen = r['btLen']
        seqid = np.float32(r['identical']) / np.float32(alnlen)
        qcov = np.float32(min(qL, max(r['qStart'], r['qEnd'])) - min(r['qStart'], r['qEnd']) + 1) / np.float32(qL)
        if not (r['evalue'] <= 10.0 and qcov >= np.float32(0.8) and alnlen >= 30):
            return None
    bits = int(orc.bitscore(r['score']) + 0.5)
    cig, st, c = [], 'M', 0
    for ch in r['backtrace']:
        if ch != st:
            cig.append('%d%s' % (c, st))
            st, c = ch, 1
        else:
            c += 1
    cig.append('%d%s' % (c, st))
    return '%d\t%d\t%d\t%s\t%.3E\t%d\t%d\t%d\t%d\t%d\t%d\t%s\n' % (q, t, bits, seqid_text(seqid), r['evalue'], r['qStart'], r['qEnd'],
                                                                qL, r['tStart'], r['tEnd'], tL, ''.join(cig))


rsw = RefSW(ref, int(lens.max()), dbres)
aln_lines = []
t0 = time.time()
for q in sorted(pref):
    rsw.set_query(seqs[q])
    mine = []
    for t in pref[q]:
        r = rsw.align(seqs[t], identity=(q == t))
        l = line(q, t, r, int(lens[q]), int(lens[t]))
        if l:
            # Matcher::compareHits (Matcher.h:157-168): the order Alignment::run writes a query's lines in
            mine.append(((r['evalue'], -int(orc.bitscore(r['score']) + 0.5), int(lens[t]), t), l))
    mine.sort(key=lambda x: x[0])
    aln_lines += [l for _, l in mine]
print('alignment lines', len(aln_lines), round(time.time() - t0, 1), 's', flush=True)
for name, lines, want in (('config1_pref.tsv.gz', pref_lines, '8109a70bdea70ee10e0dbd27ba6b7e37'),
                          ('config1_aln.tsv.gz', aln_lines, '2e917f0e9782e8a7412c7360aa7bf1b4')):
    # the fixture keeps DB order (entries by query key, lines as the module writes them); the recorded md5 is of the sorted lines
    md5 = hashlib.md5(b''.join(sorted(l.encode() for l in lines))).hexdigest()
    assert md5 == want, (name, md5, want)
    with gzip.GzipFile(os.path.join(GOLD, name), 'wb', 9, mtime=0) as f:
        f.write(''.join(lines).encode())
    print(name, len(lines), 'lines, md5 of the sorted lines', md5)
