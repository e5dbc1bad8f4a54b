cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "" pixo_amd/ab_*.so; do
  name=${lib:-default}
  for wl in ${AB_WORKLOADS:-c2 c2_444 c3}; do
    PIXO_BENCH_ABLATION=${lib:+1} PIXO_HIP_LIB=${lib:+$PWD/$lib} python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('%-28s %-8s value %9.0f Mpx/s  kernel %7.2f us  frac %.3f' % ('$name', '$wl', d['value'], r['kernel_us_avg'], r['frac']))"
  done
done
