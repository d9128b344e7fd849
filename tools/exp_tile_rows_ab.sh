# round 6's tile-height rule for scalings near 1 against round 5's (SMOL_TILE_ROWS_R5=1, tuning build), outputs in rotation, arenas placed
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
f() { python tests/bench_configs_checked.py --configs $1 --iters 12 --placement far 2>/dev/null | python -c "
import sys,json
print(' '.join('%s=%.1f(%.3f)%s' % (d['config'].split(':')[5] if d['config'].startswith('G:') else d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"; }
C=R1.03,R1.1,R1.2,R1.35,R1.5,R1.9,U1.1,U1.37,U1.5,U2,S4K91ODD,S4K109ODD,G:2:3840:2160:2:3955:2224:8,G:2:3840:2160:2:4953:2786:8,G:2:3840:2160:2:5414:3045:6,G:2:3840:2160:2:6950:3909:4,G:2:1920:1080:2:3000:1688:12,G:2:7680:4320:2:7000:3938:2
for p in new old new old; do if [ $p = old ]; then echo "round 5's rule   $(SMOL_TILE_ROWS_R5=1 f $C)"; else echo "round 6's rule   $(f $C)"; fi; done
