R=$PWD
python tools/bench_configs.py --iters 20 --warmup 3 --check --configs C1,C2,C3,C4,C5,C5mix,BOX,R1.5,R3,R6,PHOTO,PNG,P16,SRGB,P16UP,SRGBUP,BOX128 > gpurun_out/configs_final.jsonl 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_stats -- python3 $R/tools/bench_configs.py --iters 10 --warmup 2 --configs C1,C2,C3,C4,C5,C5mix,BOX,R3,PHOTO,PNG,P16,SRGB > $R/gpurun_out/prof3_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof3_fetch -- python3 $R/tools/bench_configs.py --iters 3 --warmup 1 --configs C3,C4,BOX,R3 > $R/gpurun_out/prof3_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof3_write -- python3 $R/tools/bench_configs.py --iters 3 --warmup 1 --configs C3,C4,BOX,R3 > $R/gpurun_out/prof3_write.log 2>&1
cd $R; cut -c1-18,150-300 gpurun_out/configs_final.jsonl; cut -c1-110 gpurun_out/prof3_stats/*/*kernel_stats.csv | head -20
