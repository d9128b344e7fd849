cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
CF=SRGBUP,SRGBPNGUP,C3,P16UP
bash tools/ab_once.sh $CF libsmolscale_amd_before.so libsmolscale_amd_tuning.so libsmolscale_amd_before.so libsmolscale_amd_tuning.so
python tests/bench_configs_checked.py --configs SRGBUP --iters 20 --opaque 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('opaque', d['config'], d['step_us'], d['frac_of_8TBps'], d.get('bit_exact_first_rows_vs_oracle'))"
