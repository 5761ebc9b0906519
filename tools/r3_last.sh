set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3l_gpu_all.log 2>&1 || true
tail -2 gpurun_out/r3l_gpu_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err
python -c "
import json; j=json.loads(open('gpurun_out/r3l_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['steps'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'], j['host_enqueue_ms_per_step'])"
