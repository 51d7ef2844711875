for v in ${HGYM_AB_MODES:-HGYM_ASYNC_SAVE=1 HGYM_ASYNC_SAVE=0 HGYM_ASYNC_SAVE=1 HGYM_ASYNC_SAVE=0}; do
  env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pmc --configs logging 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); c=d['configs'][0]; print('$v', 'head %.4g' % d['value'], {k:(round(v,2) if isinstance(v,float) else v) for k,v in c.items() if k in ('value','ms_per_step','collection_ms','ppo_update_ms','checkpoint_ms_total','final_checkpoint_wait_ms_after_learn','steps','warmup')})"
done
