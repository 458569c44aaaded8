"""The DB boundary on the CPU (no device calls): `sdgpu createsetdb` reproduces the files the reference's createsetdb
leaves for the regression input byte for byte, the DB reader takes the reference's split data files, and the glue
modules between `align` and `clusterhits` reproduce the reference binary's intermediate DBs on its own alignment DB
(fixture written from the real reference classes, tools/make_golden_alndb.py; md5s recorded from the reference
binary's run)."""
import hashlib
import os

import pytest

from dbutil import (SDGPU, write_db, read_db, flat_lines_from_gz, entries_by_first_column, sorted_md5, sdgpu, example_fasta)

# md5 of the files `spacedust createsetdb examples/*.faa genome tmp` writes (reference binary, survey container)
SETDB_MD5 = {
    'genome': 'e7419320ddf755d7a442068f3c6fd1cb', 'genome.index': 'd4ee4a52af874bff5be7288a66c9ef22',
    'genome.lookup': '9d4c45a834309ca4e21b1426fe53ea69', 'genome_h': 'a15153d9cf7c380d46d780f90f8a811b',
    'genome_set_to_member': '2ea6fdb1c0c448f95cc236046bddaa95', 'genome_member_to_set': '5267d2085c116d65569faee3725ac15f',
    'genome_set_size': 'd7dc61ac6b82c8f8fb4a4debde94454f', 'genome.source': 'daed820a3f2f3eb7f523bd697579ec59',
}


@pytest.fixture(scope='module')
def setdb(tmp_path_factory):
    tmp = tmp_path_factory.mktemp('setdb')
    fa = example_fasta(tmp)
    sdgpu('createsetdb', fa[0], fa[1], tmp / 'genome', tmp / 'tmp', '-v', '0')
    return tmp


def test_binary_exists_and_lists_modules():
    assert os.path.exists(SDGPU), 'spacedust_amd/sdgpu is built by spacedust_amd.build (build())'
    p = sdgpu('--help')
    for m in ('prefilter', 'align', 'clusterhits', 'clustersearch', 'combinehits', 'summarizeresults'):
        assert m in p.stdout


def test_createsetdb_layout_equals_reference_files(setdb):
    for f, want in SETDB_MD5.items():
        assert hashlib.md5(open(setdb / f, 'rb').read()).hexdigest() == want, f
    assert open(setdb / 'genome.dbtype', 'rb').read() == b'\x00\x00\x00\x00'
    assert open(setdb / 'genome_h.dbtype', 'rb').read() == b'\x0c\x00\x00\x00'
    assert open(setdb / 'genome_set_size.dbtype', 'rb').read() == b'\x0c\x00\x00\x00'
    assert open(setdb / 'genome_member_to_set.dbtype', 'rb').read() == b'\x05\x00\x00\x00'


def test_unrecognised_flag_and_unsupported_values_fail_loudly(setdb):
    p = sdgpu('prefilter', setdb / 'genome', setdb / 'genome', setdb / 'p', '--no-such-flag', '1', check=False)
    assert p.returncode != 0 and 'Unrecognized parameter' in p.stderr
    p = sdgpu('prefilter', setdb / 'genome', setdb / 'genome', setdb / 'p', '--exact-kmer-matching', '1', check=False)
    assert p.returncode != 0 and 'not supported' in p.stderr
    p = sdgpu('align', setdb / 'genome', setdb / 'genome', setdb / 'p', setdb / 'a', '--gap-open', 'aa:9,nucl:5', check=False)
    assert p.returncode != 0


def test_glue_chain_on_reference_alignment_db(setdb):
    """prefixid -> besthitbyset -> mergeresultsbyset -> combinehits (R/data/clustersearch.sh:121-140) on the alignment DB of
    the regression input, read from 8 split data files as the reference's 8-thread DBWriter leaves them."""
    aln = flat_lines_from_gz('config1_aln.tsv.gz')
    assert sorted_md5(aln) == '2e917f0e9782e8a7412c7360aa7bf1b4'
    t = setdb
    write_db(str(t / 'result'), entries_by_first_column(aln, 5898), 5, splits=8)
    sdgpu('prefixid', t / 'result', t / 'flat.tsv', '--tsv', '--threads', '1')
    assert sorted_md5(open(t / 'flat.tsv').readlines()) == '2e917f0e9782e8a7412c7360aa7bf1b4'
    sdgpu('prefixid', t / 'result', t / 'result_prefixed', '--threads', '8', '-v', '3')
    sdgpu('besthitbyset', t / 'genome', t / 'genome', t / 'result_prefixed', t / 'aggregate', '--simple-best-hit', '1',
          '--suboptimal-hits', '0', '--threads', '8', '--compressed', '0', '-v', '3')
    sdgpu('mergeresultsbyset', t / 'genome_set_to_member', t / 'aggregate', t / 'aggregate_merged', '--threads', '8', '-v', '3')
    sdgpu('combinehits', t / 'genome', t / 'genome', t / 'aggregate_merged', t / 'matches', t / 'tmp', '--alpha', '1',
          '--aggregation-mode', '0', '--filter-self-match', '1', '--threads', '8', '--compressed', '0', '-v', '3')
    # md5 of the reference binary's own intermediate DBs (flattened with prefixid --tsv, sorted)
    want = {'result_prefixed': ('7498de551f136d45f3b554adf5e48928', 15065), 'aggregate': ('04ed8b4ff4fb9406fc5f12367c355e11', 7204),
            'aggregate_merged': ('3928c57ffb541c45b018793b90d4edf1', 7204)}
    for db, (md5, n) in want.items():
        sdgpu('prefixid', t / db, t / (db + '.flat'), '--tsv')
        lines = open(t / (db + '.flat')).readlines()
        assert (len(lines), sorted_md5(lines)) == (n, md5), db
    # matches / matches_h: the reference numbers these entries per worker thread (Aggregation.cpp:121,147), so the key
    # column is compared away
    for db, md5, n in (('matches', '48fdf88348f8a3139fae8ac02a6d97d7', 1283), ('matches_h', '9b9935454c88b6fdbcccd70068638296', 2)):
        sdgpu('prefixid', t / db, t / (db + '.flat'), '--tsv')
        lines = open(t / (db + '.flat')).readlines()
        assert (len(lines), sorted_md5(lines, drop_first_column=True)) == (n, md5), db
    hdr = read_db(str(t / 'matches_h'))
    assert hdr[0] == b'0\t1\t4319\t1579\t732\t0.000E+00\n' and hdr[1] == b'1\t0\t1579\t4319\t551\t0.000E+00\n'
    assert open(t / 'matches.dbtype', 'rb').read() == b'\x05\x00\x00\x00'
    assert open(t / 'aggregate_merged.dbtype', 'rb').read() == b'\x05\x00\x02\x00'   # DBTYPE_EXTENDED_INDEX_NEED_SRC


def test_createindex_file_equals_the_reference_index_file(setdb):
    """`sdgpu createindex` writes NAME.idx in PrefilteringIndexReader's layout.  The per-key md5s below are
    those of the file `spacedust createindex genome tmp -s 5.7` (reference binary) wrote for the same DB: ENTRIES (packed
    6-byte records), ENTRIESOFFSETS (size_t), the masked SEQINDEXDATA, both extended score matrices, the serialized sequence
    and header DBs, META -- everything except the matrix file text (regenerated, same numbers) and the GENERATOR string."""
    sdgpu('createindex', setdb / 'genome', setdb / 'tmp', '-s', '5.7', '--threads', '8', '-v', '0')
    want = {0: 'a31d4efb05ad73d6ea3c0dfc5b323b99', 1: 'e0640873b3744141c00e26025c036a19', 3: '9b3162d5c618b8a0b552b4fdaf529545',
            4: 'a63cdf5a33acc9817ce5edb755308e85', 5: 'e0bffb64aa6c5bb8c83611edbc4a1c21', 6: '3e003421cf8e31998ae50da9281c557d',
            7: 'e0bffb64aa6c5bb8c83611edbc4a1c21', 8: '3e003421cf8e31998ae50da9281c557d', 9: 'ba9915b344ea01bf28dc7f74749c57f8',
            10: '7c312c4256dbf5975f5aa75a2cdd2f45', 12: 'c3ce4b0559fd1f02e33d286c6880d80b', 13: 'ea7cd3d6ff584b2a33d084abf0de4f77',
            14: 'd8c7fe7653246e0baca25d0f256b5163', 15: 'ba6e3f11d5bf4173d0092293448bdd7f', 16: '71fd9c58f287a33ce1cd79b9f7925cd4',
            18: 'fbffa5c8b9ebb68700fe24a41863491e', 19: '94eda297ed394d8fcec45e491dd88f7d', 20: 'fbffa5c8b9ebb68700fe24a41863491e',
            21: '94eda297ed394d8fcec45e491dd88f7d', 23: '93b885adfe0da089cdf634904fd59f71'}
    idx = {int(l.split()[0]): (int(l.split()[1]), int(l.split()[2])) for l in open(str(setdb / 'genome.idx') + '.index')}
    assert sorted(idx) == sorted(list(want) + [2, 22])
    assert open(str(setdb / 'genome.idx') + '.dbtype', 'rb').read() == b'\x09\x00\x00\x00'
    with open(setdb / 'genome.idx', 'rb') as f:
        for k, (off, length) in sorted(idx.items()):
            assert off % 4096 == 0 or k in (7, 8, 20, 21)          # DBWriter::alignToPageSize
            if k not in want:
                continue
            f.seek(off)
            h, left = hashlib.md5(), length
            while left > 0:
                b = f.read(min(left, 1 << 24))
                h.update(b)
                left -= len(b)
            assert h.hexdigest() == want[k], k
    # the wide index form (>= 2^32 entries; forced here) goes to the file as the same absolute size_t offsets: identical bytes
    import subprocess
    from dbutil import SDGPU
    first = open(setdb / 'genome.idx', 'rb').read()
    os.remove(setdb / 'genome.idx')
    p = subprocess.run([SDGPU, 'createindex', str(setdb / 'genome'), str(setdb / 'tmp'), '-s', '5.7', '--threads', '8', '-v', '0'],
                       env=dict(os.environ, SD_INDEX_WIDE='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    second = open(setdb / 'genome.idx', 'rb').read()
    k_off, k_len = idx[1]   # ENTRIESOFFSETS and everything index-related; the GENERATOR / matrix text keys are compared above
    assert len(first) == len(second) and first[k_off:k_off + k_len] == second[k_off:k_off + k_len]
    assert hashlib.md5(second[idx[0][0]:idx[0][0] + idx[0][1]]).hexdigest() == want[0]
    os.remove(setdb / 'genome.idx')
