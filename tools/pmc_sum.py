"""Sums rocprofv3 --pmc counter_collection csv per kernel name prefix: python tools/pmc_sum.py <csv> <substr>"""
import csv
import sys
from collections import defaultdict

tot = defaultdict(float)
n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot):
    print(f"    {k:34s} {tot[k]:16.0f}")
