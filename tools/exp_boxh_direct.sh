# boxes of more than ~9 600 pixels per column: k_boxh straight from memory (round 6) against the backstop k_general<true>
# (tuning build: SMOL_NO_BOXH_DIRECT=1 = before). profiles/r06_boxh_direct.txt
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 10 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('   %-34s %-44s %9.1f us  %.3f of 8 TB/s %s' % (d['config'], d['kernel'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!'))"; }
C=G:0:65535:2048:0:1:32:4,G:0:65535:2048:0:4:2048:4,G:8:65535:1024:8:2:1024:4:1,G:4:40000:1024:4:3:90:4,G:1:20001:4096:1:2:4096:4:1
echo "straight from memory (shipped)"; f $C
echo "SMOL_NO_BOXH_DIRECT=1 (round 5: backstops)"; SMOL_NO_BOXH_DIRECT=1 f $C
