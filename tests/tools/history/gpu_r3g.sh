# round 3, call H: hidden-block weight gradients transposed, 16-byte partial-row accesses
set -x
mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3h
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bg or 128 or background or image" ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for v in "" "--ws-no-tail"; do
timeout 200 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline $v > $O/bench_background$v.json 2>&1 < /dev/null; tail -1 $O/bench_background$v.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'])"
done
timeout 100 python tests/tools/phase_profile.py background split 2 > $O/phases_background_flags2.txt 2>&1; tail -19 $O/phases_background_flags2.txt
true
