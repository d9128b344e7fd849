# round 6's rules for k_pow2's workgroup height (2:1 of small frames) and k_up's tile height into 24-bit rows against round 5's, interleaved
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=C1,G:0:1920:1080:0:960:540:40,G:0:4096:4096:0:2048:2048:6,U4RGB,G:8:1920:1080:8:5760:3240:6,G:0:1920:1080:8:7680:4320:4,G:8:960:540:8:7680:4320:4,G:4:1920:1080:9:4993:2809:6
for p in new old new old; do if [ $p = old ]; then echo "round 5's rules   $(SMOL_POW2_BY16=1 SMOL_UP_ROWS_R5=1 f $C)"; else echo "round 6's rules   $(f $C)"; fi; done
