#!/usr/bin/env python3
"""BASELINE configs[3] at its target size on one GPU: `sdgpu clustersearch Q T out tmp --num-iterations 3` with T = P synthetic
proteomes (default 1 000: 3 * 10^6 proteins) and Q = the first q of them (default 20) as their own set DB (createsetdb from one FASTA
file per proteome), timed: the iterations run in memory (csrc/cli/sd_mod_iter.cpp, no DB between the modules).  Then, untimed:
  * the same command with --keep-tmp 1 -- the module chain over DB files -- on the first `parity_sets` query proteomes: its TSV must be
    the head of the timed run's TSV, byte for byte (entries are ordered by query set, clusters numbered in that order);
  * the DBs that chain left behind are checked on a sample of queries against the reference's own classes run on this machine
    (oracle/_ref/libsdref*.so), iteration by iteration (M/data/workflow/blastpgp.sh:73-133):
    prefilter   profile_{k-1} vs T       == QueryMatcher driven with the DBTYPE_HMM_PROFILE Sequence, row for row
    align       on the subtracted rows   == Matcher::getSWResult with the profile query: targets, coordinates, backtraces, E-values
    result2profile                       == MultipleAlignment / MsaFilter / PSSMCalculator, byte for byte (profile_0 from the
                                            sequence search's alignments, profile_1 from the merged ones)

  python tools/iter3_scale.py [P [q [sample]]]        on the GPU box; prints one JSON line
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

def _write_fasta(ps, s, out_dir, lut):
    """one proteome = one FASTA file with Prodigal-style headers (accession # start # end # strand)"""
    a, e = int(ps.set_start[s]), int(ps.set_start[s + 1])
    text = lut[ps.residues[int(ps.offsets[a]):int(ps.offsets[e])]].tobytes()
    base = int(ps.offsets[a])
    parts = []
    for i in range(a, e):
        o0, o1 = int(ps.offsets[i]) - base, int(ps.offsets[i + 1]) - base
        start = 1000 + 2000 * (i - a)
        parts.append(b'>p%05d_%d # %d # %d # %d\n' % (s, i - a, start, start + 3 * (o1 - o0), 1 if ps.strand[i] else -1))
        parts.append(text[o0:o1])
        parts.append(b'\n')
    path = os.path.join(out_dir, 'p%05d.faa' % s)
    with open(path, 'wb') as f:
        f.write(b''.join(parts))
    return path


def _compress(bt):
    out, state, count = [], 'M', 0
    for ch in bt:
        if ch != state:
            out.append('%d%s' % (count, state))
            state, count = ch, 1
        else:
            count += 1
    out.append('%d%s' % (count, state))
    return ''.join(out)


def _entries(path, keys):
    """the DB entries of `keys` only (the data files at this size are large): dict key -> list of tab-split lines"""
    want = set(int(k) for k in keys)
    loc = {}
    for line in open(path + '.index'):
        k, o, l = line.split()
        if int(k) in want:
            loc[int(k)] = (int(o), int(l))
    out = {}
    with open(path, 'rb') as f:
        for k, (o, l) in loc.items():
            f.seek(o)
            out[k] = f.read(l - 1)
    return out


def _lines(db):
    return {k: [l.split('\t') for l in v.decode().split('\n') if l] for k, v in db.items()}


def run(P=1000, q_sets=20, sample=32, threads=None, log=print, keep_dir=None, parity_sets=2):
    from dbutil import SDGPU
    from spacedust_amd import api
    from spacedust_amd.cpus import effective_cpus
    from spacedust_amd.synth import make_proteomes, ALPHABET
    from oracle.pyoracle import Ref, RefSW, RefResult2Profile, ref_available, ref_r2p_available
    threads = threads or effective_cpus()
    work = keep_dir or tempfile.mkdtemp(prefix='sd_iter3_')
    os.makedirs(work, exist_ok=True)
    out = dict(target_proteomes=P, query_proteomes=q_sets, threads=threads)
    try:
        t0 = time.time()
        ps = make_proteomes(P, genes_per_proteome=3000, seed=0x5ED0 + 2)
        out['generate_s'] = time.time() - t0
        fa_dir = os.path.join(work, 'fa')
        os.makedirs(fa_dir, exist_ok=True)
        t0 = time.time()
        lut0 = np.frombuffer(ALPHABET.encode(), np.uint8)
        files = [_write_fasta(ps, s_, fa_dir, lut0) for s_ in range(P)]   # (no worker processes: the caller may hold a HIP context)
        out['fasta_s'] = time.time() - t0

        def sdgpu(*a):
            r = subprocess.run([SDGPU] + [str(x) for x in a], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError('sdgpu %s failed: %s' % (a[0], (r.stderr or r.stdout)[-400:]))
            return r

        T, Q = os.path.join(work, 'T'), os.path.join(work, 'Q')
        t0 = time.time()
        sdgpu('createsetdb', *files, T, os.path.join(work, 'tmpT'), '-v', '0')
        sdgpu('createsetdb', *files[:q_sets], Q, os.path.join(work, 'tmpQ'), '-v', '0')
        parity_sets = min(parity_sets, q_sets)
        Q2 = os.path.join(work, 'Q2')
        sdgpu('createsetdb', *files[:parity_sets], Q2, os.path.join(work, 'tmpQ2'), '-v', '0')
        out['createsetdb_s'] = time.time() - t0
        shutil.rmtree(fa_dir)
        verbose = os.environ.get('SD_ITER3_VERBOSE') == '1'
        # ---- timed: the iterations in memory
        t0 = time.time()
        os.environ['SD_ITER_PROFILE'] = '1'   # per-kernel event times of the workers' contexts, one JSON object on stderr
        r = sdgpu('clustersearch', Q, T, os.path.join(work, 'iter3.tsv'), os.path.join(work, 'tmpm'), '--num-iterations', '3', '--threads', threads,
                  '-v', '3')
        wall = time.time() - t0
        os.environ.pop('SD_ITER_PROFILE', None)
        if verbose:
            log(r.stdout[-6000:])
            log(r.stderr[-12000:])
        nq = int(ps.set_start[q_sets])
        tsv = open(os.path.join(work, 'iter3.tsv')).readlines()
        out.update(wall_s=wall, queries=nq, genome_pairs=q_sets * P, genome_pairs_per_s=q_sets * P / wall, how='in memory (sd_mod_iter.cpp)',
                   hit_lines=sum(1 for l in tsv if l.startswith('>')), cluster_lines=sum(1 for l in tsv if l.startswith('#')),
                   stages=[l.strip() for l in r.stdout.splitlines() if l.startswith('iteration ') or l.startswith('in-memory iterations')],
                   files_between_modules=sorted(os.listdir(os.path.join(work, 'tmpm'))) if os.path.isdir(os.path.join(work, 'tmpm')) else [])
        for l in r.stderr.splitlines():
            if l.startswith('[iter profile] '):
                try:
                    kern = json.loads(l[len('[iter profile] '):])
                    kern = {k: v for k, v in kern.items() if not k.startswith('host:') and not k.startswith('stat.')}   # (host scopes and counters are not kernel times)
                    out['kernel_ms'] = {k: round(v[0], 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])[:14]}
                    out['kernel_ms_total'] = round(sum(v[0] for v in kern.values()), 1)
                    by_stage = dict(prefilter=sum(v[0] for k, v in kern.items() if k.startswith('prefilter_')),
                                    sw_score=sum(v[0] for k, v in kern.items() if k.startswith('sw_score')),
                                    sw_traceback=sum(v[0] for k, v in kern.items() if k.startswith('sw_traceback')),
                                    result2profile=sum(v[0] for k, v in kern.items() if k.startswith('r2p_')))
                    out['kernel_ms_by_stage'] = {k: round(v, 1) for k, v in by_stage.items()}
                    # the prefilter stage's roofline over the three iterations: SURVEY.md 8(d) bytes of the measured counters over the
                    # event-timed duration of its kernels (several workers share the device: the sum of their kernels' times)
                    st = next((l for l in r.stdout.splitlines() if l.startswith('prefilter stats:')), None)
                    if st and by_stage['prefilter'] > 0:
                        w = st.split()
                        K, H, Cn, L, R = (int(w[i]) for i in (3, 5, 7, 9, 11))
                        alg = 16 * K + 6 * H + 15 * Cn + L + 21 * R   # (+ 10 x the prefilter rows, not counted here)
                        out['roofline'] = dict(bound='hbm', stage='prefilter (sequence + two profile searches)', algorithmic_bytes=alg,
                                               kernel_ms=round(by_stage['prefilter'], 1), achieved=alg / by_stage['prefilter'] / 1e6, peak=8000.0,
                                               unit='GB/s', frac=alg / by_stage['prefilter'] / 1e6 / 8000.0, kmers=K, index_hits=H,
                                               wall_frac=alg / wall / 1e9 / 8000.0)
                except ValueError:
                    pass
        log('clustersearch --num-iterations 3 (in memory):', round(wall, 1), 's,', out['hit_lines'], 'hits in', out['cluster_lines'], 'clusters')
        # ---- untimed: the module chain over DB files on the first query proteomes; same TSV
        tmp = os.path.join(work, 'tmp')
        t0 = time.time()
        r = sdgpu('clustersearch', Q2, T, os.path.join(work, 'iter3_chain.tsv'), tmp, '--num-iterations', '3', '--keep-tmp', '1', '--threads', threads,
                  '-v', '3' if verbose else '0')
        wall2 = time.time() - t0
        if verbose:
            log(r.stdout[-6000:])
            log(r.stderr[-12000:])   # SD_DEBUG_TIMING=1: the modules' lap times
        chain = open(os.path.join(work, 'iter3_chain.tsv')).readlines()
        out['module_chain'] = dict(query_proteomes=parity_sets, wall_s=wall2, genome_pairs_per_s=parity_sets * P / wall2, tsv_lines=len(chain),
                                   tsv_equals_head_of_in_memory_tsv=bool(len(chain) > 0 and tsv[:len(chain)] == chain))
        log('module chain on', parity_sets, 'query proteomes:', round(wall2, 1), 's; TSV = head of the in-memory TSV:', out['module_chain']['tsv_equals_head_of_in_memory_tsv'])
        if not out['module_chain']['tsv_equals_head_of_in_memory_tsv']:   # where they part
            first = next((i for i, (x, y) in enumerate(zip(chain, tsv)) if x != y), min(len(chain), len(tsv)))
            sc, sm = set(chain), set(tsv[:len(chain)])
            out['module_chain'].update(first_difference_at_line=first, chain_line=chain[first][:300] if first < len(chain) else None,
                                       in_memory_line=tsv[first][:300] if first < len(tsv) else None, lines_only_in_chain=len(sc - sm),
                                       lines_only_in_memory_head=len(sm - sc),
                                       context_chain=[l[:200] for l in chain[max(0, first - 2):first + 3]], context_in_memory=[l[:200] for l in tsv[max(0, first - 2):first + 3]])
            def entry_lines(lines, names):   # every line of the (query set, target set) entry `names`, clusters in file order
                got, on = [], False
                for l in lines:
                    if l.startswith('#'):
                        on = l.split('\t')[1:3] == names
                    if on:
                        got.append(l[:160])
                return got
            ref_line = chain[first] if chain[first].startswith('#') else next((chain[i] for i in range(first, -1, -1) if chain[i].startswith('#')), '#')
            names = ref_line.split('\t')[1:3]
            ec, em = entry_lines(chain, names), entry_lines(tsv[:len(chain) + 100000], names)
            log('entry', names, 'chain:', len(ec), 'lines, in memory:', len(em), 'lines; same set of > lines:', sorted(l for l in ec if l[0] == '>') == sorted(l for l in em if l[0] == '>'))
            hc, hm = set(l for l in ec if l[0] == '>'), set(l for l in em if l[0] == '>')
            log('ONLY IN CHAIN\n' + ''.join(sorted(hc - hm)[:12]))
            log('ONLY IN MEMORY\n' + ''.join(sorted(hm - hc)[:12]))
            genes = set(l.split('\t')[0] for l in (hc ^ hm))
            log('ALL LINES OF THOSE QUERY GENES, CHAIN\n' + ''.join(l for l in ec if l.split('\t')[0] in genes)[:3000])
            log('ALL LINES OF THOSE QUERY GENES, IN MEMORY\n' + ''.join(l for l in em if l.split('\t')[0] in genes)[:3000])
            # the chain's merged alignment DB lines of the first such query gene against that target set
            try:
                for g in sorted(genes):
                    qkey = int(g[1:].split('_')[1]) + int(ps.set_start[int(g[2:7])])
                    tset = int(names[1][1:6])
                    for dbn in ('search/aln_0', 'search/aln_tmp_1', 'search/aln_tmp_2', 'result'):
                        ent = _lines(_entries(os.path.join(tmp, dbn), [qkey])).get(qkey, [])
                        log('CHAIN %s lines of query key %d against target set %d (%d lines in the entry)' % (dbn, qkey, tset, len(ent)))
                        for w in ent:
                            if int(ps.set_start[tset]) <= int(w[0]) < int(ps.set_start[tset + 1]):
                                log('   ', '\t'.join(w)[:200])
            except Exception as e_:
                log('no result DB lines:', repr(e_))
            log('first difference', json.dumps({k: out['module_chain'][k] for k in ('first_difference_at_line', 'chain_line', 'in_memory_line', 'lines_only_in_chain', 'lines_only_in_memory_head', 'context_chain', 'context_in_memory')}, indent=1))
        nq = int(ps.set_start[parity_sets])
        if not (ref_available() and ref_r2p_available()):
            out['parity_check'] = dict(queries=0, note='oracle/_ref/libsdref*.so did not travel')
            return out

        # ---------------- sampled parity against the reference classes
        t0 = time.time()
        S = os.path.join(tmp, 'search')
        rng = np.random.default_rng(17)
        lens = ps.lengths()
        # queries that have something to align in the profile iterations are the interesting ones: sample among those
        idx1 = [int(l.split()[0]) for l in open(os.path.join(S, 'aln_tmp_1.index')) if int(l.split()[2]) > 1]
        pool_q = np.array(sorted(idx1)) if idx1 else np.arange(nq)
        qs = np.sort(rng.choice(pool_q, min(sample, len(pool_q)), replace=False)).tolist()
        lut = np.frombuffer(ALPHABET.encode(), np.uint8)
        blob = lut[ps.residues].tobytes()
        seq_of = lambda t: blob[int(ps.offsets[t]):int(ps.offsets[t + 1])].decode()
        ref = Ref(6)
        rix = ref.index(blob, ps.offsets, kmer_thr=0, threads=threads)     # profile searches index every k-mer (Prefiltering.cpp:525-527)
        out['reference_index_s'] = time.time() - t0
        host = api.Host()
        thr = host.profile_kmer_threshold(5.7, 6)
        rpf = rix.prefilter_profile(int(lens.max()) + 10, thr, max_hits=300)
        rsw = RefSW(ref, int(lens.max()) + 10, int(ps.offsets[-1]))
        r2p = RefResult2Profile()
        bad = dict(prefilter=0, align=0, profile=0)
        n = dict(prefilter_rows=0, alignments=0, profiles=0)
        ref_s = dict(prefilter=0.0, align=0.0, profile=0.0)   # seconds inside the reference classes (one thread)
        # profile_0 from the sequence search's alignments
        prof0 = _entries(os.path.join(S, 'profile_0'), qs)
        aln0 = _lines(_entries(os.path.join(S, 'aln_0'), qs))

        for q in qs:
            rows = aln0.get(q, [])
            et, eq, ets, bts = [], [], [], []
            for w in rows:
                if float(w[3]) < 0.001:
                    et.append(int(w[0])); eq.append(int(w[4])); ets.append(int(w[7])); bts.append(api.uncompress_cigar(w[10]))
            t1 = time.time()
            got = r2p.profile(seq_of(q), [seq_of(t) for t in et], eq, ets, bts)
            ref_s['profile'] += time.time() - t1
            bad['profile'] += got != prof0[q]
            n['profiles'] += 1
        for step in (1, 2):
            last = step == 2
            prof = _entries(os.path.join(S, 'profile_%d' % (step - 1)), qs)
            got_pf = _lines(_entries(os.path.join(S, 'pref_tmp_%d' % step), qs))
            todo = _lines(_entries(os.path.join(S, 'pref_%d' % step), qs))
            aln = _lines(_entries(os.path.join(S, 'aln_tmp_%d' % step), qs))
            for q in qs:
                t1 = time.time()
                ids, sc, dg, _ = rpf.query(prof[q])
                ref_s['prefilter'] += time.time() - t1
                keep = (lens[ids].astype(np.float32) / np.float32(lens[q])) >= np.float32(0.8)   # Util::canBeCovered, --cov-mode 2
                want = [(int(t), int(s_), int(np.int16(np.uint16(d)))) for t, s_, d in zip(ids[keep], sc[keep], dg[keep])]
                mine = [(int(w[0]), int(w[1]), int(w[2])) for w in got_pf.get(q, [])]
                bad['prefilter'] += mine != want
                n['prefilter_rows'] += len(want)
                rows = todo.get(q, [])
                wantA = {}
                if rows:
                    t1 = time.time()
                    rsw.set_query_profile(prof[q])
                    for w in rows:
                        t = int(w[0])
                        r = rsw.align(seq_of(t), sw_mode=2, eval_thr=10.0 if last else 0.001, cov_mode=2, cov_thr=0.8)
                        if r['btLen'] > 0 and r['evalue'] <= (10.0 if last else 0.001) and len(r['backtrace']) >= 30:
                            wantA[t] = (r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], _compress(r['backtrace']), '%.3E' % r['evalue'])
                    ref_s['align'] += time.time() - t1
                mineA = {int(w[0]): (int(w[4]), int(w[5]), int(w[7]), int(w[8]), w[10], w[3]) for w in aln.get(q, [])}
                bad['align'] += mineA != wantA
                n['alignments'] += len(wantA)
            if not last:
                merged = _lines(_entries(os.path.join(S, 'aln_%d' % step), qs))
                nxt = _entries(os.path.join(S, 'profile_%d' % step), qs)
                for q in qs:
                    et, eq, ets, bts = [], [], [], []
                    for w in merged.get(q, []):
                        if float(w[3]) < 0.001:
                            et.append(int(w[0])); eq.append(int(w[4])); ets.append(int(w[7])); bts.append(api.uncompress_cigar(w[10]))
                    t1 = time.time()
                    got = r2p.profile(None, [seq_of(t) for t in et], eq, ets, bts, centre_profile=prof[q])
                    ref_s['profile'] += time.time() - t1
                    bad['profile'] += got != nxt[q]
                    n['profiles'] += 1
        out['parity_check'] = dict(queries=len(qs), prefilter_rows=n['prefilter_rows'], prefilter_queries_mismatching=bad['prefilter'],
                                   alignments=n['alignments'], alignment_queries_mismatching=bad['align'], profiles=n['profiles'],
                                   profiles_mismatching=bad['profile'], seconds=time.time() - t0,
                                   against='oracle/_ref/libsdref.so + libsdref_r2p.so (the reference classes, this machine)')
        log('parity', out['parity_check'])
        # CPU baseline of the profile part: the reference classes, one thread, on the sampled queries (the two profile prefilters, the
        # profile alignments of both iterations, both result2profile calls; the sampled queries are those with profile alignments, so
        # this is the cost of a query that takes part).  Iteration 0 -- an ordinary sequence search with one more alignment pass -- is
        # what the main record's cpu_baseline measures; bench.py adds it.
        per_q = sum(ref_s.values()) / max(len(qs), 1)
        out['cpu_profile_iterations'] = dict(kind='reference', cores=1, queries=len(qs), seconds_per_query=per_q,
                                             seconds_by_stage={k: round(v / max(len(qs), 1), 4) for k, v in ref_s.items()},
                                             note='seconds inside QueryMatcher (profile) / Matcher::getSWResult / result2profile classes per sampled query, '
                                                  'iterations 1 and 2 + both profile computations, one thread; sequence iteration 0 not included')
        return out
    finally:
        if keep_dir is None:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:4]]
    print(json.dumps(run(*(a + [1000, 20, 32][len(a):]))))
