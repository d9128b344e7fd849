cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
echo "--- natural vs random data"
for flag in "" "--natural"; do
python tests/bench_configs_checked.py --configs SRGB,SRGBHALF,SRGBPHOTO,SRGBPNG,SRGBUP,SRGBBOX,SRGBTHUMB,BOX,R3,PHOTO --iters 20 $flag 2>/dev/null | python -c "
import sys,json
print('[$flag]', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done
echo "--- THUMB: two-pass vs single-pass k_tile"
bash tools/ab_env.sh THUMB,S1080T -- "X=1" "SMOL_BOX2_MIN=32"
