"""Per-kernel medians of the counters of one or more `rocprofv3 --kernel-trace --pmc ... --output-format csv` output directories.
  python tools/pmc_summary.py gpurun_out/pmc_x1 gpurun_out/pmc_x2 [--match k_decode]"""
import csv
import glob
import statistics
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
match = sys.argv[sys.argv.index('--match') + 1] if '--match' in sys.argv else ''
vals = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in dirs:
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            if match not in k:
                continue
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            key = (r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in vals:
    print(f'== {k}   launches {len(dur[k])}   median {statistics.median(dur[k]):.1f} us')
    v = {c: statistics.median(x) for c, x in vals[k].items()}
    for c in sorted(v):
        print(f'   {c:28s} {v[c]:16.0f}')
    wc = v.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
            if c in v:
                print(f'   {c} / SQ_WAVE_CYCLES = {v[c] / wc:.3f}')
    if 'SQ_LDS_BANK_CONFLICT' in v and v.get('SQ_LDS_IDX_ACTIVE'):
        print(f'   LDS bank conflict / idx active = {v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]:.3f}')
    if 'GRBM_GUI_ACTIVE' in v and 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        print(f'   mfma busy frac = {v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024):.3f}   clock GHz = {v["GRBM_GUI_ACTIVE"] / 8 / statistics.median(dur[k]) / 1e3:.2f}')
