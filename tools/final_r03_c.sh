# Round 3, last pass (host-side changes after tools/final_r03_b.sh: window batch arrays, tandem annotation): GPU suite, bench lines, native + assembler soaks
O=gpurun_out/final_r03c; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --config 4 --steps 3 > $O/bench_config4.json 2> $O/bench_config4.err
python bench.py --config 4 --steps 3 > $O/bench_config4_second_process.json 2> $O/bench_config4_b.err
python tools/native_soak.py ${1:-300} 2>&1 | tail -1 > $O/native_soak.json
python tests/soak/assembler_soak.py ${1:-300} 2>&1 | tail -1 > $O/assembler_soak.json
cat $O/pytest_gpu.txt; tail -c 300 $O/native_soak.json; tail -c 300 $O/assembler_soak.json
for f in bench_config4 bench_config4_second_process; do python -c "
import json
l=json.loads(open('$O/$f.json').read().strip().split(chr(10))[-1]); print('$f', round(l['value']), l['timed_s_runs'])"; done
