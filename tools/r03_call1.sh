set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_c1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_a -- python3 $R/tools/spconv_trace.py > $O/pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_b -- python3 $R/tools/spconv_trace.py > $O/pmc_b.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_a sc_ 0 > $O/pmc_a.json
python tools/pmc_summary.py $O/pmc_b sc_ 0 > $O/pmc_b.json
rm -rf $O/pmc_a $O/pmc_b
