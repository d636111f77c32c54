set -x
mkdir -p gpurun_out/wp
export TMPDIR=/tmp
O=$PWD/gpurun_out/wp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "frame_trajectory" ) > $O/pytest_frame.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -12 $O/pytest_frame.log
