mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tests/tools/phase_profile.py > gpurun_out/phases.txt 2>&1; cat gpurun_out/phases.txt
