cd $GRAFT_REPO_ROOT
H=tools/microbench/host_path
python -m pytest tests/test_hip_multi.py -m gpu -x -q 2>&1 | tail -2
for ll in 1 2 3; do echo "LINK_LANES=$ll"; for pin in 0 1; do for th in 1 2 4 16; do SMOL_HIP_LINK_LANES=$ll $H $th 8 $pin; done; done; done
