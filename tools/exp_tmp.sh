f() { python tests/bench_configs_checked.py --configs $1 --iters 20 ${@:2} 2>/dev/null | python -c "
import sys,json
for d in map(json.loads, sys.stdin): print('  ', d['config'], d['kernel'][:44], '%.1f us  %.3f' % (d['step_us'], d['frac_of_8TBps']), d.get('bit_exact_first_rows_vs_oracle'))"; }
echo default; f SRGBBILBOX,P16BILBOX,P16LBILBOX,FLATBILBOX
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
echo "SMOL_HBOX_MIN=8"; SMOL_HBOX_MIN=8 f SRGBBILBOX,P16BILBOX,P16LBILBOX
echo "SMOL_HBOX_MIN=8 SMOL_HBOX_SPLITS=2"; SMOL_HBOX_MIN=8 SMOL_HBOX_SPLITS=2 f SRGBBILBOX,P16BILBOX,P16LBILBOX
echo "SMOL_NO_HBOX=1"; SMOL_NO_HBOX=1 f SRGBBILBOX,P16BILBOX,P16LBILBOX
