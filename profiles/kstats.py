#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as a short table: python profiles/kstats.py <csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
def short(nm):
    nm = nm.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'([A-Za-z0-9_]+(<[^>]*>)?)', nm)
    return m.group(1) if m else nm
for r in rows[:n]:
    print("%-28s calls %5s  avg %10.1f us  %6s %%" % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
