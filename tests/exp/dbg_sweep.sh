# k_tile ablations (SMOL_TILE_DBG bits: 1 no taps, 2 no staging, 4 no stores) per config
CFG=${1:-C3,C4,BOX,R1.5,R3,R6,PHOTO,PNG}
for d in 0 1 2 4 7; do
  echo -n "dbg=$d "
  SMOL_TILE_DBG=$d python tools/bench_configs.py --configs $CFG --iters 30 | python -c "
import sys,json
print(' '.join('%s %.1f' % (json.loads(l)['config'], json.loads(l)['step_us']) for l in sys.stdin))"
done
