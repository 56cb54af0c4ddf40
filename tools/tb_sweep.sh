# block-size sweep of the traceback kernel on the C2 bench:  bash tools/tb_sweep.sh
for cfg in "64 1024" "128 1024" "256 1024" "128 512"; do
  set -- $cfg
  CLH_TB_SMALL_NT=$1 CLH_TB_BIG_NT=$2 timeout 100 python bench.py --workload c2 --no-cpu --steps 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value']), round(d['ms_per_step'],1), [round(l['ms'],1) for l in d['launches'][-2:]])"
done
