cd "$GRAFT_REPO_ROOT" || exit 1
p() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], d['ms_per_step'], r['frac'], r.get('launch_us'), r.get('step_us_each'))" $1 "$2"; }
for rep in 1 2; do
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs --no-e2e > gpurun_out/b1.json 2>/dev/null; p gpurun_out/b1.json "bare   "
python3 bench.py --steps 50 --warmup 5 > gpurun_out/b2.json 2>/dev/null; p gpurun_out/b2.json "full   "
python3 bench.py --steps 50 --warmup 5 --no-e2e > gpurun_out/b3.json 2>/dev/null; p gpurun_out/b3.json "no-e2e "
python3 bench.py --steps 50 --warmup 5 --no-other-configs > gpurun_out/b4.json 2>/dev/null; p gpurun_out/b4.json "no-other"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/b5.json 2>/dev/null; p gpurun_out/b5.json "no-cpu "
done
