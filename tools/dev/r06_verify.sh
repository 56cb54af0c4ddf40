# end-of-round verification at HEAD: the GPU suite, smoke(), the default bench (timed)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06z_gputests.txt 2>&1; tail -2 gpurun_out/r06z_gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-300
t0=$(date +%s); python bench.py --steps 20 --warmup 5 --detail gpurun_out/r06z_c3_bench_detail.json > gpurun_out/r06z_c3_bench.json 2> gpurun_out/r06z_c3_bench.err; echo "default bench wall: $(( $(date +%s) - t0 )) s, $(wc -c < gpurun_out/r06z_c3_bench.json) bytes"; cut -c1-300 gpurun_out/r06z_c3_bench.json
