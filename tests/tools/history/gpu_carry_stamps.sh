export TMPDIR=/tmp; timeout 300 python tests/tools/carry_stamps.py
