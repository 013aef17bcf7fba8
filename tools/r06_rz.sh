#!/bin/bash
O=gpurun_out/r06rz; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "surrounds" > $O/pytest.log 2>&1
tail -15 $O/pytest.log | cut -c1-300
