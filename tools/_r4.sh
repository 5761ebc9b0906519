set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 bash tools/pmc_conv16s.sh pmc_r2e
python tools/pmc_summary.py gpurun_out/pmc_r2e conv16s_kernel --write-json "profiles/round2_conv16s_pmc_final.txt" > gpurun_out/pmc_r2e_summary.txt
cat gpurun_out/pmc_r2e_summary.txt
cp profiles/dominant_kernel_traffic.json gpurun_out/dominant_kernel_traffic.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r2e_stats -o s -- python $R/bench.py --steps 14 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r2e_stats.log 2>&1
cd $R
timeout -k 10 600 python bench.py > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err
tail -c 1500 gpurun_out/r2e_bench.json
