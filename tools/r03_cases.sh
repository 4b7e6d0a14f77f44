cd $GRAFT_REPO_ROOT
for c in soliton jet sill stommel beach wind headline_land closed_small; do python tools/bench_case.py $c 200 2>&1 | tail -1 | cut -c1-330; done > gpurun_out/r03_other_configs.txt
echo "--- jet 2048x2048x2: single handle vs ring of one band over RCCL (bench.py --case jet [--force-bands])" >> gpurun_out/r03_other_configs.txt
python bench.py --case jet --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', round(d['ms_per_step'],4))" >> gpurun_out/r03_other_configs.txt
python bench.py --case jet --force-bands --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring of one, RCCL', round(d['ms_per_step'],4), d['config']['halo']['step_form'])" >> gpurun_out/r03_other_configs.txt
cat gpurun_out/r03_other_configs.txt
timeout -k 10 600 python tools/soak_bands.py 400 > gpurun_out/r03_soak_bands.txt 2>&1; tail -16 gpurun_out/r03_soak_bands.txt | cut -c1-200
