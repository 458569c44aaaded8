"""sums the per-kernel table of tools/isolated_times.sh by stage (usage: python tools/iso_sum.py isolated_kernel_times.txt)"""
import re, sys
tot = {}
for line in open(sys.argv[1]):
    m = re.match(r'^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s', line)
    if not m or line.startswith('kernel'):
        continue
    name, ms = m.group(1), float(m.group(3))
    if 'sw_score' in name: g = 'sw_score'
    elif 'traceback' in name or name.startswith('k_bt') : g = 'sw_traceback'
    elif 'clusterhits' in name: g = 'clusterhits'
    elif any(x in name for x in ('kmers', 'kp_', 'join_', 'hot_filter', 'segment_match', 'partition_hits', 'bucket_', 'score_diag', 'keep_max', 'select_hits',
                                 'coarse_', 'gather_hits', 'col_prefix', 'scan_', 'small_scan', 'query_', 'compact_kernel', 'cand_stats', 'bin_count', 'flag_to')): g = 'prefilter'
    else: g = 'other'
    tot[g] = tot.get(g, 0.0) + ms
print({k: round(v, 1) for k, v in sorted(tot.items())}, 'sum', round(sum(tot.values()), 1))
