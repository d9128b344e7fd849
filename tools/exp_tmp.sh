cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/gpu_tests_final.txt
cat gpurun_out/gpu_tests_final.txt
python3 tools/fuzz/gpu_fuzz.py 91 30000 2>&1 | tail -2 | cut -c1-300
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 20 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'][:24], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
A=G:8:1920:1080:8:4993:2809:4,G:8:1920:1080:8:5120:2880:4,G:8:1920:1080:8:5760:3240:4,U4RGB,G:0:1920:1080:8:4993:2809:4,G:0:1920:1080:8:5120:2880:4,G:0:1920:1080:8:5763:3241:4,G:0:1920:1080:8:7680:4320:4
for rep in 1 2; do
echo "k_tile   $(SMOL_NO_UP_P3=1 f $A)"
echo "shipped  $(f $A)"
done
