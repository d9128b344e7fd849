# knobs of the streaming kernels that rounds 1-3 tuned into an output arena the Infinity Cache kept, once more with the outputs in rotation
# and the arenas placed (tuning build; profiles/r06_retune_rot.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 15 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
echo "k_pow2 shipped (16 rows per workgroup; 8 for rows that share lines)   $(f C1,C2,C5,C2CROP4,C2ODD)"
for by in 2 4 8 16; do echo "k_pow2 BY=$by                                       $(SMOL_POW2_BY=$by f C1,C2,C5,C2CROP4,C2ODD)"; done
echo "k_hspan shipped (1 row per workgroup)   $(f C4)"
for r in 2 4 8; do echo "k_hspan rows=$r                         $(SMOL_HSPAN_ROWS=$r f C4)"; done
echo "k_up shipped   $(f C3,U2.6,U3,U8,U4RGB)"
for r in 4 6 8 12 16; do echo "k_up rows=$r    $(SMOL_UP_ROWS=$r f C3,U2.6,U3,U8,U4RGB)"; done
