set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_vx; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_pointpillar_path.py tests/test_gpu_bench_paths.py -m gpu -x -q -k "voxelize or pointpillar_kitti_bs16" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for c in uniform ring; do
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --resident >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --flush --iters 30 >> $O/vx.log 2>&1
  timeout -k 10 120 python tools/vx_bench.py --algos 3 --cloud $c --resident --flush --iters 30 >> $O/vx.log 2>&1
done
grep algo $O/vx.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_full -- python3 $R/tools/vx_bench.py --algos 3 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_res -- python3 $R/tools/vx_bench.py --algos 3 --resident > /dev/null 2>&1
cd $R
for d in kt_full kt_res; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo $d; head -4 $f; done
