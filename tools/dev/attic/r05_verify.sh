# end-of-round verification at HEAD: the GPU suite, smoke(), the default bench (timed), kernel stats + counters of the C3 step
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05e_gputests.txt 2>&1; tail -2 gpurun_out/r05e_gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s); python bench.py --steps 20 --warmup 5 --detail gpurun_out/r05e_c3_bench_detail.json > gpurun_out/r05e_c3_bench.json 2> gpurun_out/r05e_c3_bench.err; echo "default bench wall: $(( $(date +%s) - t0 )) s"; wc -c gpurun_out/r05e_c3_bench.json; cut -c1-300 gpurun_out/r05e_c3_bench.json
bash tools/pmc_run.sh r05e_c3 --steps 10 --warmup 2 | tail -3
