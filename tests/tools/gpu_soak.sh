# soak: the GPU suite five times in a row on one box (races in the LDS overlays / barriers of the new kernels would show as flakes)
set -x
mkdir -p gpurun_out/soak
export TMPDIR=/tmp
O=$PWD/gpurun_out/soak
for i in 1 2 3 4 5; do
( timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_$i.log 2>&1 < /dev/null; echo "run $i rc=$?"; grep -n "passed\|failed" $O/pytest_$i.log | tail -1
done
