cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
python tools/fuzz/gpu_fuzz.py 47 4000 2>&1 | tail -2
CF=SRGBHALF,SRGBPHOTO,SRGBPNG
bash tools/ab_once.sh $CF libsmolscale_amd_before.so libsmolscale_amd_tuning.so libsmolscale_amd_before.so libsmolscale_amd_tuning.so
