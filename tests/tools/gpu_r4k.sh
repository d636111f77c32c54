# round 3, call 4K: HBM-side traffic of the configs[4] shape (256 objects x 256 rays, hidden 64, step_main_wp<2>), bf16 and f32 weights
set -x
mkdir -p gpurun_out/r4k gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4k
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc/$tag
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tests/tools/run_steps.py stress_256x64 20 > $O/pmc_$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
import json
out={k:{c:sum(v)/len(v) for c,v in cs.items()} for k,cs in acc.items()}
json.dump(out,open('gpurun_out/r4k/pmc_counters_stress.json','w'),indent=1)
for k,cs in out.items():
    if 'FETCH_SIZE' in cs: print(k, 'fetch x2 MB', 2*cs['FETCH_SIZE']/1024, 'write MB', cs.get('WRITE_SIZE',0)/1024, 'mfma busy', cs.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*cs.get('SQ_WAVE_CYCLES',1)), 'hit', cs.get('TCC_HIT_sum',0)/(cs.get('TCC_HIT_sum',0)+cs.get('TCC_MISS_sum',1)))
PY
rm -rf gpurun_out/pmc
true
