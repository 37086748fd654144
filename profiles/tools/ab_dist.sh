for i in 1 2 3; do for e in 1 0; do
  LORADS_AR_PLAIN=$e LORADS_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu --no-extra 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain=$e', round(d['value'],1), round(d['ms_per_step'],4))"
done; done
python bench.py --no-cpu --no-extra 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single', round(d['value'],1), round(d['ms_per_step'],4))"
