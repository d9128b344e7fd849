for b in 4096 8192 12288 16384 24576; do for r in 8 16 32; do
  echo -n "budget=$b rows=$r "
  SMOL_TILE_LDS_BUDGET=$b SMOL_TILE_ROWS=$r python tools/bench_configs.py --configs ${1:-C4} --iters 30 | python -c "
import sys,json
print(' '.join('%s %.1f' % (json.loads(l)['config'], json.loads(l)['step_us']) for l in sys.stdin))"
done; done
