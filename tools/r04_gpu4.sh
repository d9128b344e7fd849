cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
python -m pytest tests -m gpu -x -q > gpurun_out/r04d/pytest.txt 2>&1; tail -3 gpurun_out/r04d/pytest.txt
tools/repro/run_pageable_heap.sh 60 20 > gpurun_out/r04d/repro.txt 2>&1; grep -v "^  \[" gpurun_out/r04d/repro.txt | tail -8
python bench.py > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err; python - <<'PY'
import json
l = json.loads(open('gpurun_out/r04d/bench.json').read().strip().splitlines()[-1])
print(l['value'], l['roofline']['frac'], l['roofline'].get('frac_profiled'), json.dumps(l.get('e2e'))[:3000])
PY
