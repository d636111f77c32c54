mkdir -p gpurun_out/pmc_bg; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_bg/$tag -o p -- python $R/tests/tools/run_steps.py background 20 > $R/gpurun_out/pmc_bg/$tag.log 2>&1
  echo "$tag rc=$?"
done
cd $R; python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_bg/*/p_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        n=row['Kernel_Name']
        k='gen' if 'step_main_gen' in n else 'fin' if 'finalize' in n else None
        if k: acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k,cs in acc.items():
    print(k,{c:round(sum(v)/len(v),1) for c,v in sorted(cs.items())})
PY
