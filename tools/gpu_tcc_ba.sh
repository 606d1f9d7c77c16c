cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CORB_BA_NO_GRAPH=1
timeout 500 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCC_EA0_RDREQ_sum TCC_BUSY_avr -d /tmp/tcc -o tcc -- python tools/ba_store_scale.py 6250 > /dev/null 2> /tmp/tcc.log
python tools/rocprof_summary.py /tmp/tcc/tcc_results.db gpurun_out/r03_tcc.txt > /dev/null || tail -5 /tmp/tcc.log
grep -E "ba_pcg_spmv|ba_pcg_step_big|ba_schur_mfma" gpurun_out/r03_tcc.txt | cut -c1-140
