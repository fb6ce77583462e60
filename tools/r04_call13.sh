R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout -k 10 400 bash tools/vx_pmc_collect.sh > $O/vx_pmc.log 2>&1; tail -5 $O/vx_pmc.log | cut -c1-300
cp gpurun_out/voxelize_pmc.json $O/voxelize_pmc.json; mkdir -p profiles/r04; cp gpurun_out/voxelize_pmc.json profiles/r04/voxelize_pmc.json
rm -rf gpurun_out/vx_pmc_full_fetch gpurun_out/vx_pmc_full_write gpurun_out/vx_pmc_resident_fetch gpurun_out/vx_pmc_resident_write
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest3.log 2>&1 || { tail -60 $O/pytest3.log; exit 1; }
tail -2 $O/pytest3.log
timeout -k 10 600 python bench.py --stages > $O/bench3.log 2>&1 || { tail -30 $O/bench3.log; exit 1; }
grep stages $O/bench3.log
tail -1 $O/bench3.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); e=d['extra']; r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['kernel_sum_us'], r['traffic'], r['traffic_ratio'], r['timed_path']['traffic_ratio_vs_own'])
print({k:e[k] for k in ('second_kitti','pvrcnn_kitti','second_multihead_nuscenes','spconv_gemm','h2d_inclusive_frames_per_s','spconv_forward_ms') if k in e})
"
