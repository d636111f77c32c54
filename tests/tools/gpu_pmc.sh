set -x
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
grep -ciE "mfma" $R/gpurun_out/pmc/counters_list.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tests/tools/run_steps.py ${PMC_CONFIG:-replica_room0_vmap} 40 ${PMC_WEIGHTS:-f32} > $R/gpurun_out/pmc/$tag.log 2>&1
  echo "$tag rc=$?"
done
cd $R; find gpurun_out/pmc -name "*.csv" | head -30
