# k_up's tile height by magnification and output type, outputs in rotation, arenas placed (tuning build: SMOL_UP_ROWS; profiles/r06_retune_rot.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
# 32-bit premultiplied out: x5 x6 x8 x12 ; 24-bit out: RGB8->RGB8 x3 x4 x8, RGBA->RGB8 x4
A=G:0:1536:864:0:7680:4320:4,G:0:1280:720:0:7680:4320:4,U8,G:0:640:360:0:7680:4320:4,C3
B=G:8:1920:1080:8:5760:3240:6,U4RGB,G:8:960:540:8:7680:4320:4,G:0:1920:1080:8:7680:4320:4,G:4:1920:1080:9:4993:2809:6
echo "shipped     $(f $A,$B)"
for r in 5 6 7 8 10; do echo "rows=$r      $(SMOL_UP_ROWS=$r f $A)"; done
for r in 8 12 16 20 24 32; do echo "rows=$r     $(SMOL_UP_ROWS=$r f $B)"; done
# k_pow2 2:1 at large frames
echo "k_pow2 2:1 4096->2048 x6, 1920x1080->960x540 x40: shipped $(f G:0:4096:4096:0:2048:2048:6,G:0:1920:1080:0:960:540:40)  BY=4 $(SMOL_POW2_BY=4 f G:0:4096:4096:0:2048:2048:6,G:0:1920:1080:0:960:540:40)  BY=8 $(SMOL_POW2_BY=8 f G:0:4096:4096:0:2048:2048:6,G:0:1920:1080:0:960:540:40)"
