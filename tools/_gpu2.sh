mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02_pytest2.log
timeout 900 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
tail -8 gpurun_out/r02_pytest2.log; tail -5 gpurun_out/r02_bench_a.err; wc -c gpurun_out/r02_bench_a.json
