"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/ab_stage.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/ab_stage.py
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 6400000 profiles/r02_pmc_hbm_traffic.csv "tools/ab_stage.py: one 100000-ray batch"

Writes the compact csv bench.py reads (bytes per sample per kernel; median over launches), stamped with the hash of the
kernel sources it was taken from so that bench.py can tell whether the numbers belong to the build it is timing.
Counter units and the gfx950 caveat follow MI355X_MICROARCH.md (HBM): the counters are in KB; FETCH_SIZE tallies a wide
coalesced 128-B request at 64 B, so streaming reads are under-counted by up to 2x; scattered 4-16 B gathers are uncalibrated."""
import csv
import glob
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def medians(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return {k: statistics.median(v) for k, v in out.items()}


def main():
    fetch_dir, write_dir, samples, out, what = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    from bench import source_hash
    fe, wr = medians(fetch_dir, 'FETCH_SIZE'), medians(write_dir, 'WRITE_SIZE')
    with open(out, 'w') as f:
        f.write(f'# source_hash={source_hash()} rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on {what}; '
                f'{samples:.0f} samples per launch; median over launches; counters in KB as reported\n')
        f.write('# gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of wide coalesced streams (x2 for those); '
                'scattered 4-16 B gathers are uncalibrated\n')
        f.write('kernel,FETCH_SIZE_KB,WRITE_SIZE_KB,fetch_bytes_per_sample,write_bytes_per_sample\n')
        for k in fe:
            if not k.startswith(('k_', 'void k_')):
                continue
            a, b = fe[k], wr.get(k, 0.0)
            f.write('"%s",%.0f,%.0f,%.1f,%.1f\n' % (k[:60], a, b, a * 1024 / samples, b * 1024 / samples))
    print(open(out).read())


if __name__ == '__main__':
    main()
