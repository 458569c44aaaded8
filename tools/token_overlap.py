"""Share of a file's tokens that lie in runs of >= N tokens also present (verbatim) in a reference file; comments stripped.
usage: token_overlap.py mine.cpp ref1.cpp [ref2.cpp ...] [--n 12]"""
import re
import sys


def tokens(path):
    s = open(path, errors='replace').read()
    s = re.sub(r'/\*.*?\*/', ' ', s, flags=re.S)
    s = re.sub(r'//[^\n]*', ' ', s)
    return re.findall(r'[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*[fFeE]?[-+]?\d*|\S', s)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    n = int(sys.argv[sys.argv.index('--n') + 1]) if '--n' in sys.argv else 12
    mine = tokens(args[0])
    for ref in args[1:]:
        rt = tokens(ref)
        grams = set(tuple(rt[i:i + n]) for i in range(len(rt) - n + 1))
        covered = [False] * len(mine)
        for i in range(len(mine) - n + 1):
            if tuple(mine[i:i + n]) in grams:
                for j in range(i, i + n):
                    covered[j] = True
        print('%s vs %s: %.1f %% of %d tokens in shared runs of >= %d' % (args[0], ref, 100.0 * sum(covered) / max(len(mine), 1), len(mine), n))


if __name__ == '__main__':
    main()
