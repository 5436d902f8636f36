"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/parse_pmc.py <dir> [counter]."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
per = OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:70], r['Counter_Name'])
    per.setdefault(k, []).append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
for (k, c), v in per.items():
    v.sort()
    vals = [x for _, x in v]
    print('%-72s %-12s n=%3d mean=%.6g' % (k, c, len(vals), sum(vals) / len(vals)))
    if 'conv3d_k3_kernel<ConvCfg<1, 32' in k or 'apply' in k:
        print('    per dispatch:', ' '.join('%.6g' % x for x in vals))
