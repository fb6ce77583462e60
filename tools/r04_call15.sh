set -e
mkdir -p gpurun_out/c15
timeout -k 10 300 python -m pytest tests/test_gpu_pointpillar_path.py -x -q -m gpu -k "nms" > gpurun_out/c15/nms_tests.log 2>&1
timeout -k 10 300 python bench.py --stages > gpurun_out/c15/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c15/prof -o pp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $GRAFT_REPO_ROOT/gpurun_out/c15/prof.log 2>&1
