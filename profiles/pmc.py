#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 counter_collection.csv files: python profiles/pmc.py <csv>... [--k name,...]"""
import csv, collections, sys
files=[a for a in sys.argv[1:] if not a.startswith('--k=')]
keys=[a[4:].split(',') for a in sys.argv[1:] if a.startswith('--k=')]
keys=keys[0] if keys else ['k_fine_area','k_flatten_items<true>','k_flatten_items<false>']
for f in files:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in keys:
            if key in r['Kernel_Name']:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(k,{c:f"{sum(x)/len(x):.4g}" for c,x in sorted(v.items())})
