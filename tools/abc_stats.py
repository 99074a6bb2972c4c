"""mean / median / min of the lines tools/env_abc.sh and tools/ab_many.sh print, per configuration:  tools/abc_stats.py <log>"""
import collections
import re
import sys

d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+(\w+)\s+step ([0-9.]+) ms\s+kernel ([0-9.]+)", l)
    if m:
        d[m.group(1).strip()].append((float(m.group(3)), float(m.group(4))))
for k, v in d.items():
    s, kk = sorted(x[0] for x in v), sorted(x[1] for x in v)
    print("%-44s n %2d  step mean %.4f median %.4f min %.4f | kernel mean %.4f median %.4f min %.4f"
          % (k, len(v), sum(s) / len(s), s[len(s) // 2], s[0], sum(kk) / len(kk), kk[len(kk) // 2], kk[0]))
