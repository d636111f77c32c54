set -x
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=$PWD/gpurun_out/final
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
