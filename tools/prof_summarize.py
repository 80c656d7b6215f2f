"""Condense the rocprofv3 output of tools/prof_round.sh: kernel-stats table + per-kernel PMC averages."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
for f in glob.glob(os.path.join(root, 'stats', '**', '*kernel_stats.csv'), recursive=True):
    print('== kernel stats', os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print('  %-70s calls %5s  avg %10.1f us  total %5.1f %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
for d in sorted(glob.glob(os.path.join(root, 'pmc*')) + glob.glob(os.path.join(root, '[fw]'))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('== PMC', os.path.basename(d), '(mean per dispatch)')
    for k, cs in acc.items():
        if any(x in k for x in ('gemm_bf16x3', 'gemm_lif_sparse', 'compress_planes', 'permute_planes', 'gemm_mx', 'conv3x3', 'lif_scan', 'li_heads', 'encode', 'rate', 'det_payload')):
            print('  %-60s' % k, '  '.join('%s=%.4g (n=%d)' % (c, sum(v) / len(v), len(v)) for c, v in sorted(cs.items())))
