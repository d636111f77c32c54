# round 3, call 4G: hardware counters of the background step's final kernels (three-tile single-round step_main_ws), all groups
set -x
mkdir -p gpurun_out/r4g gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4g
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc/$tag
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tests/tools/run_steps.py background 40 > $O/pmc_$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
python tests/tools/pmc_summary.py > $O/pmc_counters_background.json 2>$O/pmc_summary.err; python -c "
import json; j=json.load(open('$O/pmc_counters_background.json')); print(json.dumps(j.get('step_main_ws'),indent=0)[:1500]); print(j['_notes'])"
rm -rf gpurun_out/pmc
true
