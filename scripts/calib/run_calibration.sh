#!/bin/bash
# On the GPU box: FETCH_SIZE and WRITE_SIZE (separate --pmc passes, --kernel-trace only) of the calibration kernels, against their
# known byte counts.  -> gpurun_out/<tag>_counter_calibration.txt        usage: scripts/calib/run_calibration.sh <tag>
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
B=$GRAFT_REPO_ROOT/scripts/calib/counter_calibration
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -o $B scripts/calib/counter_calibration.hip
: > $O/${TAG}_counter_calibration.txt
for N in 16384 65536 262144; do
  $B $N 24 > /tmp/cal_known_$N.txt
  for PMC in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal_$PMC
    (cd /tmp && rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/cal_$PMC -o cal -- $B $N 24) > /tmp/cal_$PMC.log 2>&1
    cp $(find /tmp/cal_$PMC -name "*counter_collection.csv" | head -1) /tmp/cal_${PMC}_$N.csv
  done
  python - $N >> $O/${TAG}_counter_calibration.txt <<'PY'
import csv, sys, collections
n = int(sys.argv[1])
known = {}
for l in open('/tmp/cal_known_%d.txt' % n):
    p = l.split()
    if p and p[0] == 'known_bytes':
        known[p[1]] = (int(p[3]), int(p[5]))
vals = {}
for pmc in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open('/tmp/cal_%s_%d.csv' % (pmc, n))):
        k = r['Kernel_Name'].split('(')[0]
        if r['Counter_Name'] == pmc:
            agg[k].append(float(r['Counter_Value']))
    vals[pmc] = {k: sum(v[4:]) / max(1, len(v[4:])) for k, v in agg.items()}      # (the first launches touch cold pages)
print('n_agents %d  (counter values as rocprofv3 reports them: KB per launch, mean of launches 5..24)' % n)
print('%-18s %12s %12s | %12s %10s | %12s %10s' % ('kernel', 'read B', 'written B', 'FETCH_SIZE', 'x1024/read', 'WRITE_SIZE', 'x1024/wr'))
for k, (rd, wr) in known.items():
    f, w = vals['FETCH_SIZE'].get(k, float('nan')), vals['WRITE_SIZE'].get(k, float('nan'))
    print('%-18s %12d %12d | %12.1f %10s | %12.1f %10s' % (k, rd, wr, f, '%.3f' % (f * 1024 / rd) if rd else '-', w, '%.3f' % (w * 1024 / wr) if wr else '-'))
print()
PY
done
cat $O/${TAG}_counter_calibration.txt
