R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py -x -q > $O/wino_test.log 2>&1 || { tail -40 $O/wino_test.log; exit 1; }
tail -2 $O/wino_test.log
timeout -k 10 500 python tools/wino_probe.py "0 0:WINO_SCHED=0 0:WINO_SCHED=1 2 4 8 14" 16,64,248,216 16,128,124,108 16,256,62,54 16,128,200,176 > $O/wino_probe3.log 2>&1; cat $O/wino_probe3.log
