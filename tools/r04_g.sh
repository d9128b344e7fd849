cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
bash tools/ab_once.sh SRGBBOX,BOX128,WIDEBOX,SRGBBOXBIL,P16BOXBIL libsmolscale_amd_before.so libsmolscale_amd_tuning.so libsmolscale_amd_before.so libsmolscale_amd_tuning.so
