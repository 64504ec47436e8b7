F="--no-cpu-baseline --no-roofline --no-reuse-mode --no-lbfgs"
one() { python bench.py --batch $1 --steps $2 --warmup 30 $F 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('B', $1, 'ms', j['ms_per_step'], 'patches/s', j['value'])"; }
one 128 100 && one 256 100 && one 512 50 || exit 1
echo "--- two concurrent B=128"
(one 128 600 > gpurun_out/c1.txt) & P1=$!
(one 128 600 > gpurun_out/c2.txt) & P2=$!
wait $P1 && wait $P2 && cat gpurun_out/c1.txt gpurun_out/c2.txt
