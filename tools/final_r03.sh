# Round-3 evidence in one GPU call: the GPU suite, the profile set, the default bench line, the 2-rank launch, the soaks.
# usage (from the container): gpurun --timeout 4500 -- "PLAT_COMMIT=$(git rev-parse --short HEAD) bash tools/final_r03.sh [seconds per soak]"
T=${1:-450}
O=gpurun_out/final_r03; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
bash tools/profile_round3.sh > $O/profile.log 2>&1
python bench.py > $O/bench_line.json 2> $O/bench_line.err
python bench.py --gpus 2 --steps 100 --no-extras > $O/bench_2ranks.json 2> $O/bench_2ranks.err
python bench.py --gpus 2 --config 4 --regions 1024 > $O/bench_c4_2ranks.json 2> $O/bench_c4_2ranks.err
bash tools/run_soaks_r03.sh $T > $O/soaks.log 2>&1
cat $O/pytest_gpu.txt; tail -c 600 $O/bench_line.json; tail -n 12 $O/soaks.log
