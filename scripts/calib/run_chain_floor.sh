#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace averages of the chain-floor kernels.  -> gpurun_out/<tag>_chain_floor.txt
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
B=$GRAFT_REPO_ROOT/scripts/calib/chain_floor
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -o $B scripts/calib/chain_floor.hip
rm -rf /tmp/cf; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cf -o cf -- $B 300) > /tmp/cf.log 2>&1
python - $(find /tmp/cf -name "*kernel_trace.csv" | head -1) > $O/${TAG}_chain_floor.txt <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fl_" in n:
        key = n[n.index("fl_"):].split("(")[0]
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = {k: (len(v), statistics.median(v[20:]), min(v[20:]), sorted(v[20:])[int(0.9 * len(v[20:]))]) for k, v in d.items()}
print("kernel (256 workgroups x 256 threads; a writer launch in front of each)   calls  median us   min us   p90 us")
for k in sorted(rows):
    print("%-72s %6d %9.2f %8.2f %8.2f" % (k, *rows[k]))
g = lambda k: rows[k][1]
print()
print("launch floor (fl_empty)                                   %.2f us" % g("fl_empty"))
print("first global round trip + store (fl_rt<1> - fl_empty)     %.2f us" % (g("fl_rt<1>") - g("fl_empty")))
print("one more DEPENDENT round trip: (fl_rt<8> - fl_rt<1>) / 7  %.2f us;  (fl_rt<5> - fl_rt<2>) / 3  %.2f us" % ((g("fl_rt<8>") - g("fl_rt<1>")) / 7, (g("fl_rt<5>") - g("fl_rt<2>")) / 3))
print("f64 chain: 3 sqrt + sincos + atan2 (fl_f64 - fl_rt<1>)    %.2f us" % (g("fl_f64") - g("fl_rt<1>")))
print("LDS store + barrier + load + barrier: (fl_lds<5> - fl_lds<1>) / 4   %.2f us" % ((g("fl_lds<5>") - g("fl_lds<1>")) / 4))
print("36 KB from global memory into LDS + barrier (fl_stage - fl_empty)   %.2f us" % (g("fl_stage") - g("fl_empty")))
PY
cat $O/${TAG}_chain_floor.txt
