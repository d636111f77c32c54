set -x
mkdir -p gpurun_out/r6g
O=$PWD/gpurun_out/r6g
timeout 900 python -m pytest tests/test_gpu_bwd6.py -m gpu -q -x -s 2>&1 | grep -v Warning | grep "six\|passed\|failed\|Error" > $O/pytest_bwd6.txt; cat $O/pytest_bwd6.txt
timeout 300 python tests/tools/grad_err_probe.py cfg2 tiny ragged 2>&1 | grep "^{" | tee $O/grad_err.jsonl
true
