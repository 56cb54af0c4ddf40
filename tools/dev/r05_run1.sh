mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05a_gputests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a_gputests.txt
tail -3 gpurun_out/r05a_gputests.txt
python bench.py --steps 20 --warmup 5 --detail gpurun_out/r05a_c3_bench_detail.json > gpurun_out/r05a_c3_bench.json 2> gpurun_out/r05a_c3_bench.err; echo "bench rc=$?"
wc -c gpurun_out/r05a_c3_bench.json; cat gpurun_out/r05a_c3_bench.json
python tools/k3_phases.py 20000 > gpurun_out/r05a_k3_phases.txt 2>&1; cat gpurun_out/r05a_k3_phases.txt
bash tools/pmc_run.sh r05a_c3 --steps 10 --warmup 2
