f() { python tools/bench_configs.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('  ', d['config'], d['kernel'][:30], '%.1f us  %.3f' % (d['step_us'], d['frac_of_8TBps']))"; }
echo default; f FLATTALL,FLATBILBOX
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for s in 64 128 256; do echo "SMOL_HBOX_SPLITS=$s"; SMOL_HBOX_SPLITS=$s f FLATTALL,FLATBILBOX; done
echo "SMOL_NO_HBOX=1"; SMOL_NO_HBOX=1 f FLATTALL,FLATBILBOX
