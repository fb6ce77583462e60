R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_wino.py tests/test_gpu_pointpillar_path.py -k "wino or golden" -x -q > $O/wino_test.log 2>&1 || { tail -40 $O/wino_test.log; exit 1; }
tail -2 $O/wino_test.log
timeout -k 10 300 python tools/wino_bench.py > $O/wino_bench.log 2>&1 || { tail -20 $O/wino_bench.log; exit 1; }
cat $O/wino_bench.log | cut -c1-200
timeout -k 10 600 python bench.py --stages --no-cpu-baseline > $O/bench_wino2.log 2>&1 || { tail -20 $O/bench_wino2.log; exit 1; }
grep stages $O/bench_wino2.log
tail -1 $O/bench_wino2.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['extra']
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_sum_us'])
print({k:e[k] for k in ('second_kitti','pvrcnn_kitti','second_multihead_nuscenes','spconv_gemm','h2d_inclusive_frames_per_s') if k in e})
"
