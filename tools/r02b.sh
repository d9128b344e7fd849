mkdir -p gpurun_out/r02b; O=gpurun_out/r02b
python -m pytest tests/test_hip_multi.py -m gpu -x -q 2>&1 | tail -3
python bench.py --config c4 --steps 50 > $O/c4.json 2>>$O/err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02b/c4.json")); print(d["value"], d.get("e2e"))
PY
