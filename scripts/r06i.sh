#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
run8() { local t0=$(date +%s.%N); python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --full-line --envs 512 --batch 512 --replay 8192 --env-only-iters 5 --gpus $1 --same-device --dist-backend gloo --p2p-check-every 8 --no-exchange-leg $2 > /tmp/o8.json 2> /tmp/e8.txt; local rc=$?; local t1=$(date +%s.%N); echo "world $1 $2 rc $rc wall $(python -c "print(round($t1-$t0,1))") s"; python -c "
import json
try:
    d=json.loads(open('/tmp/o8.json').read().strip().splitlines()[-1]); print('ms_per_pass', d['ms_per_pass'], 'selftest_ms', d.get('exchange_selftest_ms'), 'exchange', d.get('exchange'), 'ident', d.get('ranks_bit_identical'))
except Exception as e: print('no line', e); print(open('/tmp/e8.txt').read()[-1500:])
"; }
run8 8; run8 8; run8 4; run8 8 "--config 4"
