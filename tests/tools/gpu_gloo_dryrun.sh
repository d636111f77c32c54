# bench.py's N > 1 code path with N gloo ranks on the one GPU of the box (timings mean nothing; it exercises the plumbing: flag
# reduction, per-rank times, device assertions skipped for gloo, background legs - ray-sharded, owner-computes, all-reduce alone - under
# the watchdog, rank 0's single JSON line)
N=${1:-8}
mkdir -p gpurun_out/gloo
VMAP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --steps 40 --warmup 5 --preheat-ms 30 --profile-reps 20 > gpurun_out/gloo/bench_gloo_$N.out 2> gpurun_out/gloo/bench_gloo_$N.err
echo rc=$?
grep "^{" gpurun_out/gloo/bench_gloo_$N.out | tail -1 > gpurun_out/gloo/bench_gloo_$N.json
python - <<PY
import json
j=json.load(open("gpurun_out/gloo/bench_gloo_$N.json"))
print("n_gpus", j["n_gpus"], "value %.1f M" % (j["value"]/1e6), "per-rank ms", [round(x,4) for x in j["world"]["ms_per_step_per_rank"]])
wb=j["with_background"]; print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if "ms_per_step" in kk or "us" in kk}) for k,v in wb.items() if k in ("ray_sharded","owner_computes","error")})
PY
