mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tests/tools/multistream_exp.py > gpurun_out/multistream.json 2> gpurun_out/multistream.err; echo rc=$?; cat gpurun_out/multistream.json; tail -2 gpurun_out/multistream.err
