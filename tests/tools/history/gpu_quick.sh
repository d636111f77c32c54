set -x
mkdir -p gpurun_out/quick
export TMPDIR=/tmp
O=gpurun_out/quick
( timeout 600 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | head -c 260; echo
timeout 300 python tests/tools/phase_profile.py replica_room0_vmap split > $O/phases.txt 2>&1; tail -17 $O/phases.txt
