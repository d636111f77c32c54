set -x
mkdir -p gpurun_out/ws
export TMPDIR=/tmp
O=$PWD/gpurun_out/ws
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hidden128" ) > $O/pytest_ws2.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -30 $O/pytest_ws2.log
