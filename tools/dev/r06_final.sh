# end-of-round measurements at HEAD: kernel stats + counters per workload (tools/pmc_run.sh: separate rocprofv3 --pmc passes, then --kernel-trace --stats)
mkdir -p gpurun_out
python bench.py --workload c2 --no-extra --detail gpurun_out/r06z_c2_bench_detail.json > gpurun_out/r06z_c2_bench.json 2>/dev/null; cut -c1-200 gpurun_out/r06z_c2_bench.json
bash tools/pmc_run.sh r06z_c3 --steps 10 --warmup 2 | tail -3 | cut -c1-300
bash tools/pmc_run.sh r06z_c2 --workload c2 --steps 10 --warmup 2 | tail -3 | cut -c1-300
bash tools/pmc_run.sh r06z_c4 --workload c4 --steps 4 --warmup 1 | tail -3 | cut -c1-300
bash tools/pmc_run.sh r06z_c3prod --prod-windows --steps 5 --warmup 1 | tail -3 | cut -c1-300
bash tools/pmc_script.sh r06z_collapse tools/collapse_bench.py | tail -3 | cut -c1-300
