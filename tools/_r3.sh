set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_family" > gpurun_out/t_wg.log 2>&1 || (tail -40 gpurun_out/t_wg.log; exit 1)
tail -2 gpurun_out/t_wg.log
export TMPDIR=/tmp
cd /tmp
for cs in "32 64" "64 32" "128 16"; do
  set -- $cs
  rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/wgm_${1}_s1 -o w -- python $R/tools/wgrad_micro.py $1 $2 bf16 30 > $R/gpurun_out/wgm_${1}_s1.log 2>&1
  echo "$1: $(grep 'wgrad bf16' $R/gpurun_out/wgm_${1}_s1.log) | $(grep -i 'wgrad' $R/gpurun_out/wgm_${1}_s1/w_kernel_stats.csv | awk -F, '{print $(NF-4)}' | tr '\n' ' ')"
done
cd $R
timeout -k 10 200 python tools/conv16_micro.py bf16 20 wgrad > gpurun_out/w16_micro.log 2>&1; tail -2 gpurun_out/w16_micro.log
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/b_wg.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/b_wg.log
