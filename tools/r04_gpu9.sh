cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
python tools/fuzz/gpu_fuzz.py 61 4000 2>&1 | tail -2 | cut -c1-400
bash tools/ab_env.sh PHOTO,PHOTOODD,PHOTO32ODD,PNG,R1.5RGB,R1.5U -- "SMOL_NO_LIN8=1" "X=1"
bash tools/ab_env.sh R1.5,R2.5,R3,R5,R7,R1.2 -- "X=1" "SMOL_LIN8_ALL=1"
