#!/bin/bash
# Run on the GPU box: every artifact a round commits under profiles/ (default-bench stats + PMC via profile_round.sh, the
# default bench line with its other_configs, config 3 / 4 / APF kernel stats, two ranks on one GPU with the peer exchange).
# usage: scripts/round_artifacts.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-round}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
bash scripts/profile_round.sh ${TAG} > $O/${TAG}_profile.log 2>&1
# fold the counters into profiles/summary.json HERE, so that the default bench line below carries this build's traffic
# (traffic_stale: false); the same command is repeated in the build container on the merged gpurun_out/ files
python scripts/summarize_profile.py ${TAG} envs16384_batch16384_dqn_packed > $O/${TAG}_summary_entry.json 2>&1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rm -rf /tmp/prof_c4; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs) > /tmp/c4.log 2>&1
find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_config4_kernel_stats.csv
rm -rf /tmp/prof_c3; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs) > /tmp/c3.log 2>&1
find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_config3_kernel_stats.csv
rm -rf /tmp/prof_apf; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_apf -o bench -- python $GRAFT_REPO_ROOT/bench.py --env-only --envs 32768 --uav-per-env 4 --apf --no-cpu-baseline --env-only-iters 50) > /tmp/apf.log 2>&1
find /tmp/prof_apf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_apf_envonly_kernel_stats.csv
rm -rf /tmp/prof_e65; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e65 -o bench -- python $GRAFT_REPO_ROOT/bench.py --env-only --envs 65536 --steps 10 --no-cpu-baseline) > /tmp/e65.log 2>&1
find /tmp/prof_e65 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_envonly_65536_kernel_stats.csv
# two ranks sharing this GPU (bench.py starts them itself): the peer exchange with its in-run no-exchange leg; the fault drill
python bench.py --gpus 2 --same-device --dist-backend gloo --no-cpu-baseline > $O/${TAG}_2rank_samedev_p2p.json 2> $O/${TAG}_2rank.err
python bench.py --gpus 2 --same-device --dist-backend gloo --no-cpu-baseline --inject-p2p-fault 1 --no-exchange-leg > $O/${TAG}_2rank_samedev_fault_drill.json 2>> $O/${TAG}_2rank.err
python bench.py --config 4 --gpus 2 --same-device --dist-backend gloo --envs 8192 --batch 8192 --no-cpu-baseline --steps 4 --warmup 1 > $O/${TAG}_2rank_samedev_config4_sac.json 2>> $O/${TAG}_2rank.err
python bench.py --no-cpu-baseline --no-other-configs --per 2>/dev/null > $O/${TAG}_per_bench.json
python -c "
import json
d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_pass'], d['roofline']['frac'], d['roofline_learner']['frac'])
for r in d.get('other_configs', []): print(r['baseline_config'][:40], r.get('value'), r.get('ms_per_pass'), (r.get('roofline') or {}).get('frac'), r.get('error'))
"
head -8 $O/${TAG}_kernel_stats.csv | cut -c1-150
