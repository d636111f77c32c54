set -x
mkdir -p gpurun_out/ws
export TMPDIR=/tmp
O=$PWD/gpurun_out/ws
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_all.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -12 $O/pytest_all.log
