set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_pytest1.log
timeout 600 python tools/fullsize_parity.py 512 > gpurun_out/fullsize_parity.json 2> gpurun_out/fullsize_parity.err
tail -5 gpurun_out/r02_pytest1.log; tail -3 gpurun_out/fullsize_parity.err
