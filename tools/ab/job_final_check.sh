mkdir -p gpurun_out/r4n
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4n/smoke.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4n/tests.txt 2>&1
timeout 900 python bench.py > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err
tail -2 gpurun_out/r4n/smoke.txt; tail -2 gpurun_out/r4n/tests.txt; cut -c1-300 gpurun_out/r4n/bench.json
