mkdir -p gpurun_out; export TMPDIR=/tmp
for C in replica_room0_vmap scannet0024_vmap; do
timeout 300 python tests/tools/graph_exp.py $C 2>&1 | tail -3 | tee gpurun_out/graph_exp_$C.json
done
