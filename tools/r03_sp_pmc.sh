R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_sp_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -- python3 $R/tools/spconv_trace.py > $O/pmc_$1.log 2>&1; (cd $R && python tools/pmc_summary.py $O/pmc_$1 sc_implicit 0 > $O/pmc_$1.json); rm -rf $O/pmc_$1; }
run c "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
run d "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum"
run e "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum"
run f "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS"
cd $R
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_sp_pmc/pmc_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k,v in d.items():
        if "<2, 16, 1>" in k: print(f.split("/")[-1], k[:44], v)
P
