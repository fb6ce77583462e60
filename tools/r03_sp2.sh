set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_sp; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_models_mirror.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 200 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers.log 2>&1
tail -14 $O/spconv_gemm_layers.log | cut -c1-200
LIDAR_SPCONV_REGA_KERNEL=1 timeout -k 10 200 python tools/sorted_gemm_bench.py > $O/spconv_gemm_layers_rega.log 2>&1
tail -1 $O/spconv_gemm_layers_rega.log
