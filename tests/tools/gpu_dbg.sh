timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "explode or render_only" 2>&1 | tail -5
