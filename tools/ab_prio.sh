for p in 0 1; do for wl in c2 c3 c2_444; do
  echo -n "prio_a=$p $wl: "; PIXO_HIP_PRIO_A=$p timeout 200 python bench.py --workload $wl --no-cpu-baseline --steps 400 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_avg'])"
done; done
