"""Timeline of the LAST occurrence of a kernel sequence in a rocprofv3 --kernel-trace csv: start, duration, gap to the previous kernel.
`python3 tools/trace_timeline.py <kernel_trace.csv> <name prefix of the first kernel of the sequence>`"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2]
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(first)]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print('%8.1f us  dur %6.1f  gap %6.1f  grid %8s wg %5s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Grid_Size', '?'),
                                                                  r.get('Workgroup_Size', '?'), r['Kernel_Name'][:80]))
    prev_end = e
print('span %.1f us, kernels busy %.1f us, %d launches' % ((prev_end - t0) / 1e3, busy / 1e3, i1 - i0))
