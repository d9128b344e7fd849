# the randomized GPU parity suites on the round's final binary (profiles/r06final_gpu_fuzz.txt)
echo "# randomized GPU parity runs on the round-6 FINAL binary (k_tile with ring staging, k_boxh straight from memory for spans beyond the LDS), against the oracle"
python tools/fuzz/gpu_fuzz.py 606 100000 2>&1 | tail -2
python tools/fuzz/gpu_fuzz_r05.py 607 40000 2>&1 | tail -2
python tools/fuzz/fuzz_tile_ring.py 608 3000 gpu 2>&1 | tail -1
python tools/fuzz/gpu_fuzz_opaque.py 2>&1 | tail -1
python tools/fuzz/fuzz_host_road.py 609 1500 2>&1 | tail -1
