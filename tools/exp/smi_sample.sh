# samples rocm-smi while the headline bench runs its sustained section (experiment: which limit holds the clock?)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out/smi
python3 $ROOT/bench.py --steps 2500 --warmup 20 --no-cpu-baseline --no-extra-configs --sustained-moves 0 > $ROOT/gpurun_out/smi/bench.json 2> $ROOT/gpurun_out/smi/bench.err &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | grep -E "sclk|mclk|Power|Temp|Perf" | tr '\n' ';'
  echo
  sleep 0.3
done > $ROOT/gpurun_out/smi/samples.txt 2>&1
wait $BP
rocm-smi --showclocks --showpower --showmaxpower 2>&1 | grep -E "sclk|Power|Max" | tr '\n' ';' > $ROOT/gpurun_out/smi/idle.txt
cat $ROOT/gpurun_out/smi/samples.txt; echo; cat $ROOT/gpurun_out/smi/idle.txt; echo
python3 -c "
import json; d=json.load(open('$ROOT/gpurun_out/smi/bench.json')); print(d['value'], d['ms_per_step'])"
