"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]
    key = (name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['VGPR_Count'])
    agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print('%-36s grid %7s %5s %2s vgpr %4s  n=%5d avg %8.1f us min %7.1f total %8.2f ms' % (k + (len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, sum(v) / 1e6)))
