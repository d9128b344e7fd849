cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
TABLE=C1,C2,C3,C4,C5,C5mix,BOX,C2CROP4,C2ODD,R1.2,R1.5,R1.7,R1.9,R2.5,R3,R3.5,R5,R6,R7,PHOTO,PHOTOODD,PNG,R1.5RGB,R1.5U,HDOWNVUP,THUMB,P16,SRGB,P16UP,P16HALF,SRGBUP,SRGBPNGUP,SRGBHALF,SRGBPHOTO,SRGBPNG,SRGBTHUMB,SRGBBOX,BOX128,WIDEBOX,SRGBHDVU,SRGBVCOPY,SRGBBOXBIL,P16BOXBIL
SMOL_BENCH_GPU_STATE=1 python3 tests/bench_configs_checked.py --iters 50 --warmup 3 --configs $TABLE > gpurun_out/r04g/configs.jsonl 2> gpurun_out/r04g/configs_err.txt
python - <<'PY'
import json
for l in open('gpurun_out/r04g/configs.jsonl'):
    d = json.loads(l)
    print("%-10s %-34s %8.1f us  %.3f  %s %s" % (d['config'], d['kernel'][:34], d['step_us'], d['frac_of_8TBps'], 'ok' if d.get('bit_exact_first_rows_vs_oracle') else 'MISMATCH', d.get('gpu_state', {}).get('sclk_mhz')))
PY
for c in SRGBBOX SRGBUP; do bash tools/sq_counters.sh r04g $c > /dev/null 2>&1; cat gpurun_out/r04g/sq_$c.txt | grep -v "MFMA\|UNALIGNED\|ADDR_CONF"; done
