#!/usr/bin/env python3
"""VALU calibration for bench.py's sw_valu object (profiles/r02_valu_calibration.json).

  python tools/valu_peak.py issue  OUT.json
      runs the issue micro-benchmark (tools/csrc/valu_peak.hip) on cuda:0: lane-instructions per second for the
      instruction classes of the packed Smith-Waterman score kernel, and v_pk_fma_f32 as the datasheet cross-check
      (157 TF FP32 vector = 2 flops x 2 packed lanes x 64 lanes/CU/clk).
  python tools/valu_peak.py pmc  COUNTERS.csv  BENCH.json  OUT.json
      adds instr_per_cell = 64 x SQ_INSTS_VALU summed over the score-kernel dispatches of a `rocprofv3 --pmc
      SQ_INSTS_VALU` run of `bench.py --warmup 0`  /  the forward + reverse cells the same run reports (BENCH.json = its
      JSON line).
"""
import csv
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ['v_pk_max_i16', 'v_pk_add_i16', 'v_max_i32', 'v_add_u32', 'v_pk_sub_u16', 'v_mov_b32_dpp', 'v_and_b32', 'v_pk_fma_f32']


def issue(out_path):
    L = C.CDLL(os.path.join(ROOT, 'tools', 'libvalupeak.so'))
    L.valu_peak_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_int), C.POINTER(C.c_double)]
    res = {}
    cus, clk = C.c_int(), C.c_double()
    for kind, name in enumerate(KINDS):
        best = None
        for bpc in (4, 8):   # 16 and 32 wavefronts per CU
            rate, ms = C.c_double(), C.c_double()
            rc = L.valu_peak_run(0, kind, 4000, bpc, C.byref(rate), C.byref(ms), C.byref(cus), C.byref(clk))
            if rc != 0:
                raise SystemExit('valu_peak_run failed: %d' % rc)
            if best is None or rate.value > best['lane_instr_per_s']:
                best = dict(lane_instr_per_s=rate.value, ms=ms.value, waves_per_cu=bpc * 4)
        best['lanes_per_cu_per_clk_at_reported_clock'] = best['lane_instr_per_s'] / (cus.value * clk.value * 1e9)
        res[name] = best
    sw_ops = ['v_pk_max_i16', 'v_pk_add_i16', 'v_pk_sub_u16', 'v_mov_b32_dpp', 'v_and_b32']
    out = dict(device_cus=cus.value, reported_clock_ghz=clk.value, issue=res,
               peak_lane_instr_per_s=min(res[k]['lane_instr_per_s'] for k in sw_ops),
               peak_note='slowest of the instruction classes the packed score kernel issues (%s), measured; lane-instruction = '
                         'one lane of one wavefront instruction (a packed int16 op counts once, it carries two cells)' % ', '.join(sw_ops))
    if os.path.exists(out_path):
        old = json.load(open(out_path))
        for k in ('instr_per_cell', 'instr_per_cell_source'):
            if k in old:
                out[k] = old[k]
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


def pmc(csv_path, bench_json, out_path):
    inst = 0.0
    n = 0
    for r in csv.DictReader(open(csv_path)):
        if r.get('Counter_Name') == 'SQ_INSTS_VALU' and ('sw_score_pk' in r['Kernel_Name'] or 'sw_score_kernel' in r['Kernel_Name']):
            inst += float(r['Counter_Value'])
            n += 1
    line = [x for x in open(bench_json).read().splitlines() if x.startswith('{')][-1]
    b = json.loads(line)
    # the counters of the --pmc run cover every dispatch of the process, so that run is made with --warmup 0: bench.py's
    # cell counters (timed steps only) then cover the same dispatches
    if b.get('warmup', 1) != 0:
        raise SystemExit('the --pmc run of bench.py must use --warmup 0')
    cells = b['sw_cells']['forward'] + b['sw_cells']['reverse']
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out['instr_per_cell'] = 64.0 * inst / cells
    out['instr_per_cell_source'] = ('64 x SQ_INSTS_VALU over %d sw_score / sw_score_pk dispatches (%.4g wavefront instructions) / %.4g cells '
                                    'of the same rocprofv3 --pmc run of bench.py' % (n, inst, cells))
    json.dump(out, open(out_path, 'w'), indent=1)
    print(out['instr_per_cell'], out['instr_per_cell_source'])


if __name__ == '__main__':
    if len(sys.argv) >= 3 and sys.argv[1] == 'issue':
        issue(sys.argv[2])
    elif len(sys.argv) >= 5 and sys.argv[1] == 'pmc':
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
