mkdir -p gpurun_out/r02a; O=gpurun_out/r02a
python -m pytest tests/test_bench_contract.py -m gpu -x -q > $O/contract.log 2>&1; echo "contract rc=$?" 
tail -5 $O/contract.log
for rep in 1 2; do for f in 8 16 32 48; do python bench.py --steps 50 --warmup 5 --frames-per-gpu $f --no-cpu-baseline --no-other-configs 2>>$O/err.log; done; done > $O/frames.jsonl
python bench.py --config c4 --steps 50 > $O/c4.json 2>>$O/err.log
python bench.py --config c5mix --steps 50 > $O/c5mix.json 2>>$O/err.log
python bench.py --steps 20 --warmup 3 > $O/c2_full.json 2>>$O/err.log
tail -3 $O/err.log
python - <<'PY'
import json
for l in open("gpurun_out/r02a/frames.jsonl"):
    d=json.loads(l); print(d["config"]["frames_per_step"], d["roofline"]["launch_us"], d["roofline"]["frac"], d["roofline"]["step_us_each"], d["value"])
for f in ("c4","c5mix"):
    d=json.load(open("gpurun_out/r02a/%s.json"%f)); print(f, d["value"], d["roofline"], d.get("e2e"), d["verified_vs_oracle"])
d=json.load(open("gpurun_out/r02a/c2_full.json")); print(json.dumps(d)[:3000])
PY
