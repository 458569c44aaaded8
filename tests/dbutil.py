"""Test helpers: the reference's DB file family written / read from Python (data + .index + .dbtype, optionally split
into NAME.0 .. NAME.N-1 the way the reference's multi-threaded DBWriter leaves them), and the md5 of `LC_ALL=C sort`ed lines
that SURVEY.md 8(c) quotes for the reference binary's DBs."""
import gzip
import hashlib
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SDGPU = os.path.join(ROOT, 'spacedust_amd', 'sdgpu')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def write_db(path, entries, dbtype, splits=1):
    """entries: list of (key, bytes payload without the terminator)"""
    per = (len(entries) + splits - 1) // max(splits, 1)
    index, off = [], 0
    for s in range(splits):
        name = path if splits == 1 else '%s.%d' % (path, s)
        with open(name, 'wb') as f:
            for key, payload in entries[s * per:(s + 1) * per]:
                f.write(payload + b'\0')
                index.append((key, off, len(payload) + 1))
                off += len(payload) + 1
    index.sort()
    with open(path + '.index', 'w') as f:
        for key, o, l in index:
            f.write('%d\t%d\t%d\n' % (key, o, l))
    with open(path + '.dbtype', 'wb') as f:
        f.write(struct.pack('<i', dbtype))


def read_db(path):
    """-> dict key -> payload bytes (single data file)"""
    data = open(path, 'rb').read()
    out = {}
    for line in open(path + '.index'):
        k, o, l = line.split()
        out[int(k)] = data[int(o):int(o) + int(l) - 1]
    return out


def flat_lines_from_gz(name):
    return gzip.open(os.path.join(GOLD, name), 'rt').readlines()


def entries_by_first_column(lines, n_keys):
    """flattened `key <tab> rest` lines -> DB entries (key, payload of the `rest` lines), one entry per key < n_keys"""
    by = {}
    for l in lines:
        k, rest = l.split('\t', 1)
        by.setdefault(int(k), []).append(rest)
    return [(k, ''.join(by.get(k, [])).encode()) for k in range(n_keys)]


def sorted_md5(lines, drop_first_column=False):
    if drop_first_column:
        lines = [l.split('\t', 1)[1] for l in lines]
    return hashlib.md5(b''.join(sorted(l.encode() for l in lines))).hexdigest()


def sdgpu(*args, check=True):
    p = subprocess.run([SDGPU] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if check and p.returncode != 0:
        raise AssertionError('sdgpu %s failed (%d):\n%s\n%s' % (' '.join(str(a) for a in args), p.returncode, p.stdout, p.stderr))
    return p


def example_fasta(tmp):
    out = []
    for f in ('NC_000913.faa', 'NC_000915.faa'):
        dst = os.path.join(str(tmp), f)
        with gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rb') as g, open(dst, 'wb') as o:
            o.write(g.read())
        out.append(dst)
    return out
