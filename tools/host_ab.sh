# Same-box A/B of the unchanged callers' host cost: round 5's tree (tools/ab_r05/, not in git: recreate with
#   mkdir -p tools/ab_r05 && git archive 37973af attentive_dfprior_amd tools/host_breakdown.py bench.py bench_extra.py include oracle | tar -x -C tools/ab_r05 && (cd tools/ab_r05 && bash attentive_dfprior_amd/csrc/build.sh)
# ) against this tree, alternating, three rounds; the pool's hosts are noisy (the allocation-only FLOOR of one box moved between 0.85 and 1.36 ms
# within a minute), so read the MINIMA at the end.  Mapper-shaped iteration, 5 000 rays x 64 samples, tools/host_breakdown.py.
cd $GRAFT_REPO_ROOT
[ -d tools/ab_r05 ] || { echo "tools/ab_r05 missing (see the header of tools/host_ab.sh)"; exit 1; }
T=$(mktemp)
for rep in 1 2 3; do
  ADFP_HOST_TIMING=1 python tools/ab_r05/tools/host_breakdown.py --rays 5000 --no-tracker 2>/dev/null | grep -E "free running\)|render_batch_ray \(forward\)|loss.backward|FLOOR|calls" | grep -v " 0.0 calls" | sed "s/^/round 5 | /"
  ADFP_HOST_TIMING=1 python tools/host_breakdown.py --rays 5000 --no-tracker 2>/dev/null | grep -E "free running\)|render_batch_ray \(forward\)|loss.backward|FLOOR|calls" | grep -v " 0.0 calls" | sed "s/^/round 6 | /"
done | tee $T
python - $T <<'PY'
import re, sys
rows = {}
for line in open(sys.argv[1]):
    tree = line.split('|')[0].strip()
    m = re.search(r'(render_batch_ray \(forward\)|loss\.backward\(\))\s+([\d.]+) us', line)
    if m:
        rows.setdefault((tree, m.group(1)), []).append(float(m.group(2)))
print('# per tree and section: the three occurrences per run are (host cost, synchronised, floor); minima over the three runs of the FIRST (host cost) and THIRD (floor):')
for (tree, sec), v in sorted(rows.items()):
    host, floor = v[0::3], v[2::3]
    print(f'{tree} {sec:32s} host cost min {min(host):7.1f} us   floor min {min(floor):7.1f} us   our share {min(host) - min(floor):7.1f} us')
PY
rm -f $T
