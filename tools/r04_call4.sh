R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pointpillar_path.py -k "wino or golden" -x -q > $O/wino_test.log 2>&1 || { tail -40 $O/wino_test.log; exit 1; }
tail -3 $O/wino_test.log
timeout -k 10 300 python tools/wino_bench.py > $O/wino_bench.log 2>&1 || { tail -20 $O/wino_bench.log; exit 1; }
cat $O/wino_bench.log
timeout -k 10 400 python tools/wino_probe.py "1 2 4 8 6 14 31" 16,64,248,216 16,128,124,108 16,256,62,54 > $O/wino_probe2.log 2>&1; cat $O/wino_probe2.log
