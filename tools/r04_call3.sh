R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 400 python tools/wino_probe.py "0 1 2 4 8 16 3 7 23 31" 16,64,248,216 16,128,124,108 16,256,62,54 > $O/wino_probe.log 2>&1; cat $O/wino_probe.log
timeout -k 10 400 python -m pytest tests/test_gpu_spconv.py -x -q > $O/spconv_test.log 2>&1; tail -3 $O/spconv_test.log
timeout -k 10 300 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers.log 2>&1; tail -16 $O/spconv_gemm_layers.log | cut -c1-330
timeout -k 10 300 python bench.py --mode train-ddp --steps 5 --warmup 2 > $O/train_ddp.log 2>&1; tail -2 $O/train_ddp.log | cut -c1-1200
