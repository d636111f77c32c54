mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_pipeline.py -m gpu -q -x ) > gpurun_out/pytest_pipeline.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/pytest_pipeline.log
