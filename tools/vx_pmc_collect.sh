#!/bin/bash
# HBM traffic of the voxeliser from rocprofv3 PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only:
# MI355X_MICROARCH.md "HBM"), for the full-rewrite path (algo 3) and the resident-output path (algo 4).  Run on the GPU box from
# the repo root; writes gpurun_out/vx_pmc_{full,resident}_{fetch,write}/ and gpurun_out/voxelize_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in full resident; do
  flag=""; [ "$mode" = resident ] && flag="--resident"
  for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/vx_pmc_${mode}_$(echo $c | tr A-Z a-z | cut -d_ -f1)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/vx_bench.py --algos 3 --iters 10 $flag > /dev/null 2>&1
  done
done
python3 tools/vx_pmc_json.py gpurun_out > gpurun_out/voxelize_pmc.json && cat gpurun_out/voxelize_pmc.json
