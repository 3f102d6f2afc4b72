#!/bin/bash
# Run on the GPU box: every artifact a round commits under profiles/.  For each row of the roofline table -- the default
# bench (configs[1]), env-only 65 536 / 262 144 agents, configs[2] and configs[3] -- a rocprofv3 kernel-trace + PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate passes; the default command also the SQ / MFMA sets), folded into profiles/summary.json
# HERE so that the bench lines taken afterwards carry this build's counters (traffic_stale: false); then the default bench
# line with its other_configs, the APF env-only kernel stats and the two-ranks-on-one-GPU runs.
# usage: scripts/round_artifacts.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-round}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
bash scripts/profile_round.sh ${TAG} > $O/${TAG}_profile.log 2>&1
python scripts/summarize_profile.py ${TAG} envs16384_batch16384_dqn_packed > $O/${TAG}_summary_entry.json 2>&1
PMC_SETS=traffic bash scripts/profile_round.sh ${TAG}_envonly65536 --env-only --envs 65536 > $O/${TAG}_envonly65536_profile.log 2>&1
python scripts/summarize_profile.py ${TAG}_envonly65536 envonly65536_packed >> $O/${TAG}_summary_entry.json 2>&1
PMC_SETS=traffic bash scripts/profile_round.sh ${TAG}_envonly262144 --env-only --envs 262144 > $O/${TAG}_envonly262144_profile.log 2>&1
python scripts/summarize_profile.py ${TAG}_envonly262144 envonly262144_packed >> $O/${TAG}_summary_entry.json 2>&1
PMC_SETS=traffic bash scripts/profile_round.sh ${TAG}_config3 --config 3 > $O/${TAG}_config3_profile.log 2>&1
python scripts/summarize_profile.py ${TAG}_config3 envs65536_batch65536_dueling_f16_mfma16 >> $O/${TAG}_summary_entry.json 2>&1
PMC_SETS=traffic bash scripts/profile_round.sh ${TAG}_config4 --config 4 > $O/${TAG}_config4_profile.log 2>&1
python scripts/summarize_profile.py ${TAG}_config4 envs32768_batch32768_sac_packed >> $O/${TAG}_summary_entry.json 2>&1
cp profiles/summary.json $O/${TAG}_summary.json
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cp bench_full.json $O/${TAG}_bench_full.json; cp bench_other_configs.json $O/${TAG}_bench_other_configs.json
rm -rf /tmp/prof_apf; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_apf -o bench -- python $GRAFT_REPO_ROOT/bench.py --env-only --envs 32768 --uav-per-env 4 --apf --no-cpu-baseline --steps 1) > /tmp/apf.log 2>&1
find /tmp/prof_apf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_apf_envonly_kernel_stats.csv
# two ranks sharing this GPU (bench.py starts them itself): the peer exchange with its in-run no-exchange leg; the fault drill
python bench.py --gpus 2 --same-device --dist-backend gloo --no-cpu-baseline --steps 8 --warmup 2 > $O/${TAG}_2rank_samedev_p2p.json 2> $O/${TAG}_2rank.err
python bench.py --gpus 2 --same-device --dist-backend gloo --no-cpu-baseline --inject-p2p-fault 1 --no-exchange-leg --steps 8 --warmup 2 > $O/${TAG}_2rank_samedev_fault_drill.json 2>> $O/${TAG}_2rank.err
python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every 256 --replan-count 16384 --steps 20 --warmup 4 2>/dev/null > $O/${TAG}_replan_bench.json
bash scripts/calib/run_calibration.sh ${TAG} > /dev/null 2>&1
python -c "
import json
d=json.load(open('$O/${TAG}_bench_full.json'))
r=d['roofline']
print('headline', d['value'], d['ms_per_pass'], r['frac'], r['frac_physical_stored'], r['frac_physical_counters'], r.get('traffic_stale'), d['roofline_learner']['frac'])
for r in d.get('other_configs', []):
    rr = r.get('roofline') or {}
    print(r['baseline_config'][:40], r.get('value'), r.get('ms_per_pass'), rr.get('frac'), rr.get('frac_physical_stored'), rr.get('frac_physical_counters'), rr.get('traffic_stale'), r.get('error'))
"
head -8 $O/${TAG}_kernel_stats.csv | cut -c1-150
