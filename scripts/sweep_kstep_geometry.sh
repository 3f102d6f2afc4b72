for obs in packed f16; do
for n in 32768 49152 65536 98304 131072 262144; do
for cfg in "UAVENV_TILE_STORE=0 UAVENV_COOP=0" "UAVENV_TILE_STORE=1 UAVENV_COOP=0" "UAVENV_COOP=1"; do
  if [ "$cfg" = "UAVENV_COOP=1" ] && [ $n -gt 98304 ]; then continue; fi
  r=$(env $cfg python bench.py --env-only --envs $n --steps 20 --obs-dtype $obs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f us  %.3f' % (d['k_step_ms_back_to_back']*1e3, d['frac_of_8TBs']))")
  echo "$obs $n [$cfg] $r"
done; done; done
