# Round 3, after the host-side changes that followed tools/final_r03.sh (candidate tie order, region source, thread defaults):
# the GPU suite, the bench lines and the native soak again.  usage: gpurun --timeout 2400 -- "PLAT_COMMIT=<sha> bash tools/final_r03_b.sh"
O=gpurun_out/final_r03b; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --config 4 --steps 3 > $O/bench_config4.json 2> $O/bench_config4.err
python bench.py --config 4 --steps 3 > $O/bench_config4_second_process.json 2> $O/bench_config4_b.err
python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err
python bench.py --gpus 2 --steps 100 --no-extras > $O/bench_2ranks.json 2> $O/bench_2ranks.err
python bench.py --gpus 2 --config 4 --regions 1024 > $O/bench_c4_2ranks.json 2> $O/bench_c4_2ranks.err
bash tools/profile_round3.sh c4 > $O/profile_c4.log 2>&1
python tools/native_soak.py ${1:-300} 2>&1 | tail -1 > $O/native_soak.json
cat $O/pytest_gpu.txt; tail -c 300 $O/native_soak.json
for f in bench_config4 bench_config4_second_process; do python -c "
import json
l=json.loads(open('$O/$f.json').read().strip().split(chr(10))[-1]); print('$f', round(l['value']), l['timed_s_runs'])"; done
