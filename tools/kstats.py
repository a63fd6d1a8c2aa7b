"""Prints the top rows of a rocprofv3 kernel_stats.csv with shortened kernel names: python tools/kstats.py <csv> [steps]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\(.*", "", n)[:62]
    print(f"{n:62s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step "
          f"{float(r['Percentage']):5.1f}%")
