for args in "--workload cfg3 --steps 50 --warmup 100" "--workload cfg3 --steps 200 --warmup 100" "--workload cfg3_150 --steps 50 --warmup 100" "--workload trimmed_adapters --steps 50 --warmup 100" "--workload cfg2 --steps 50 --warmup 100"; do
  python bench.py $args --no-cpu-baseline --no-tiers --no-traffic --no-steady --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', 'ms_per_step %.4f kernel %.4f gap_us %.1f total_gap_ms %.2f' % (d['ms_per_step'], r['kernel_ms'], (d['ms_per_step']-r['kernel_ms'])*1e3, (d['ms_per_step']-r['kernel_ms'])*d['steps']))"
done
