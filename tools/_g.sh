for b in 8 16 4; do for o in force_swiglu=0 force_swiglu=1; do
python3 bench.py --no-cpu-baseline --no-roofline --batch $b --steps 2 --warmup 1 --opt $o 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$b $o', d['value'], d['last_step_ms'])"
done; done
