mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tests/tools/frame_bench.py > gpurun_out/frame_bench.json 2> gpurun_out/frame_bench.err; echo rc=$?; cat gpurun_out/frame_bench.json; tail -2 gpurun_out/frame_bench.err
