#!/bin/bash
# Interleaved A/B of ENVIRONMENT settings on the tuning build (tools/build_tuning.sh) on one box:
#   bash tools/ab_env.sh <configs> [--opaque] -- "<VAR=val ...>" "<VAR=val ...>" ...      ("" = no variable)
# two passes over the settings; every row is checked against the oracle (tests/bench_configs_checked.py)
cfgs=$1; shift
extra=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do extra="$extra $1"; shift; done
shift
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
for rep in 1 2; do for setting in "$@"; do
  env $setting python tests/bench_configs_checked.py --configs $cfgs --iters 20 $extra 2>/dev/null | python -c "
import sys,json
print('[%s]' % '$setting', ' '.join('%s=%.1f(%.3f)%s' % (d['config'], d['step_us'], d['frac_of_8TBps'], '' if d.get('bit_exact_first_rows_vs_oracle') else '!!') for d in map(json.loads, sys.stdin)))"
done; done
