# round 3, call 4P: hardware counters of the hidden-256 step (imap_plumbing: step_main_ws<8>, 50 workgroups x 8 waves, one single-tile round each)
set -x
mkdir -p gpurun_out/r4p gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4p
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc/$tag
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tests/tools/run_steps.py imap_plumbing 40 > $O/pmc_$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
python tests/tools/pmc_summary.py > $O/pmc_counters_imap.json 2>$O/pmc_summary.err; python -c "
import json; j=json.load(open('$O/pmc_counters_imap.json')); print(json.dumps(j.get('step_main_ws'))); print(j['_notes'])"
rm -rf gpurun_out/pmc
true
