# end-of-round measurements at HEAD: default bench, then kernel stats + counters per workload (tools/pmc_run.sh)
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --detail gpurun_out/r05h_c3_bench_detail.json > gpurun_out/r05h_c3_bench.json 2> gpurun_out/r05h_c3_bench.err; echo "bench rc=$? $(wc -c < gpurun_out/r05h_c3_bench.json) bytes"
python bench.py --workload c2 --no-extra --detail gpurun_out/r05h_c2_bench_detail.json > gpurun_out/r05h_c2_bench.json 2>/dev/null; cut -c1-200 gpurun_out/r05h_c2_bench.json
bash tools/pmc_run.sh r05h_c3 --steps 10 --warmup 2 | tail -3
bash tools/pmc_run.sh r05h_c2 --workload c2 --steps 10 --warmup 2 | tail -3
bash tools/pmc_run.sh r05h_c4 --workload c4 --steps 4 --warmup 1 | tail -3
bash tools/pmc_run.sh r05h_c3prod --prod-windows --steps 5 --warmup 1 | tail -3
bash tools/pmc_script.sh r05h_prod tools/prod_bench.py | tail -3
