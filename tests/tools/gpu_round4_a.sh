# round 4, call A: (1) the full SQ counter set of step_main_s32 at the CURRENT 800-wave grid (the only full set
# was round 2's 960-wave grid); (2) baselines of the hidden-64 / hidden-128 shapes before this round's kernel work
set -x
mkdir -p gpurun_out/r4a gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4a
cd /tmp
rm -rf $R/gpurun_out/pmc/*
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
         "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
         "SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$tag -o p -- python $R/tests/tools/run_steps.py replica_room0_vmap 40 > $O/pmc_$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
python tests/tools/pmc_summary.py > $O/pmc_counters_step_main_s32.json 2>$O/pmc_summary.err
python -c "
import json; j=json.load(open('$O/pmc_counters_step_main_s32.json')); print(json.dumps(j.get('step_main_s32'))); print(j['_notes'])"
rm -rf gpurun_out/pmc
for cfg in "stress_256x64 bf16" "stress_256x64 f32" "stress_rank8 bf16" "stress_rank8 f32" "background f32"; do
  set -- $cfg
  timeout 300 python bench.py --config $1 --weights $2 --steps 100 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_$1_$2.json 2>$O/bench_$1_$2.err
  python -c "
import json,sys
j=json.loads(open('$O/bench_$1_$2.json').read().strip().splitlines()[-1]); r=j['roofline']
print('$1 $2', 'ms/step', round(j['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), r['launch_plan'])"
done
true
