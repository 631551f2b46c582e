# Round-3 soaks (GPU box).  usage: bash tools/run_soaks_r03.sh [seconds per soak]
T=${1:-400}
O=gpurun_out/soak_r03; mkdir -p $O
python tools/ungapped_crosscheck.py $T 70000 --bigq 2>&1 | tail -1 > $O/ungapped_bigq.json
python tools/ungapped_crosscheck.py $T 50000 2>&1 | tail -1 > $O/ungapped.json
python tools/native_soak.py $T 2>&1 | tail -3 > $O/native.json
python tests/soak/assembler_soak.py $T 2>&1 | tail -1 > $O/assembler.json
tail -n 3 $O/*.json
