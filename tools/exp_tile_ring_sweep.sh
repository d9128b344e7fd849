# k_tile's ring staging over the reference's proportional sweep (every 5th size of the 3 x 200, device-resident rows only):
# mean pps per test for the shipped two-buffer scheme and ring depths, bilinear x bilinear jobs only (tuning build;
# profiles/r06_tile_ring.txt)
export SMOLSCALE_AMD_LIB=$PWD/smolscale_amd/libsmolscale_amd_tuning.so
run() { rm -rf /tmp/sw; python tools/bench_sweep.py --no-host-row --steps 40 --out /tmp/sw 2>/dev/null | sed 's/smol-hip [0-9.e+]*, //'; }
echo "== two buffers"; run
for n in 3 4 6; do echo "== ring N=$n (classes 1+2)"; SMOL_TILE_RING=$n SMOL_TILE_RING_CLASS=3 run; done
echo "== ring N=4 (class 1 only)"; SMOL_TILE_RING=4 SMOL_TILE_RING_CLASS=1 run
echo "== two buffers"; run
