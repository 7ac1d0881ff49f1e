#!/usr/bin/env python3
"""prints (queue, stream) -> kernels seen, in order of first appearance, from a rocprofv3 kernel trace csv"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seen = collections.OrderedDict()
for r in rows:
    k = (r["Queue_Id"], r["Stream_Id"])
    seen.setdefault(k, collections.Counter())[r["Kernel_Name"].split("(")[0][:34] + " grid " + r["Grid_Size_X"]] += 1
for k, v in seen.items():
    print("queue %2s stream %2s  %s" % (k[0], k[1], dict(v)))
