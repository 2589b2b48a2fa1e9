#!/bin/bash
# Same-box A/B of the headline bench line for two builds of the library: new (in tree), old (tools/_ctrl/libcntt_hip.so), new, old.
# Prints value / ms_per_step / fwd / inv / fused kernel ms / C3 / C5 / C4 per run.   usage: tools/ab_bench_headline.sh [bench.py args]
set -e
cp concrete-ntt_amd/libcntt_hip.so /tmp/libcntt_new.so
trap 'cp /tmp/libcntt_new.so concrete-ntt_amd/libcntt_hip.so' EXIT
for which in new old new old; do
  if [ $which = new ]; then cp /tmp/libcntt_new.so concrete-ntt_amd/libcntt_hip.so; else cp tools/_ctrl/libcntt_hip.so concrete-ntt_amd/libcntt_hip.so; fi
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
c={x['config'][:2]:x for x in d['configs']}
for k in ('C3','C5'): c.setdefault(k, {'ms_per_batch': float('nan')})
c.setdefault('C4', {'fwd_ms': float('nan'), 'inv_ms': float('nan')})
print('$which', 'value %.1f M' % (d['value']/1e6), 'step %.4f ms' % d['ms_per_step'], 'fused %.4f fwd %.4f inv %.4f pw %.4f' % (r['avg_launch_ms'], r['fwd_kernel_ms'], r['inv_kernel_ms'], r['pointwise_kernel_ms']), 'sclk', r['sclk_mhz'], 'W', r['power_w'], 'C3 %.3f C5 %.3f' % (c['C3']['ms_per_batch'], c['C5']['ms_per_batch']), 'C4 %.2f/%.2f' % (c['C4']['fwd_ms'], c['C4']['inv_ms']), 'verified', d['verified'], 'unfused %.1f M' % (d['unfused_value']/1e6))"
done
