"""Turns a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the compact
summary committed under profiles/ (per-kernel calls / total / average / share)."""
import csv
import glob
import os
import sys


def main(d, out):
    f = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, 'w') as o:
        o.write(f'# source: {os.path.basename(f)} (rocprofv3 --kernel-trace --stats)\n')
        o.write('kernel,calls,total_ms,avg_us,min_us,max_us,pct\n')
        for r in rows:
            if float(r['Percentage']) < 0.05:
                continue
            o.write('"%s",%s,%.3f,%.1f,%.1f,%.1f,%s\n' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                        float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                                                        float(r['MaxNs']) / 1e3, r['Percentage']))
    print(open(out).read())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
