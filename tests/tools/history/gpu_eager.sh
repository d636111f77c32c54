mkdir -p gpurun_out; export TMPDIR=/tmp
for C in replica_room0_vmap background; do
  timeout 300 python tests/tools/torch_gpu_baseline.py $C > gpurun_out/torch_gpu_baseline_$C.json 2> gpurun_out/torch_gpu_baseline_$C.err; echo "$C rc=$?"; cat gpurun_out/torch_gpu_baseline_$C.json
done
