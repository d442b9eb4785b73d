"""Kernel timeline of the last call in a rocprofv3 kernel trace: python3 profiles/timeline.py <kernel_trace.csv> [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'at::native' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f us  +gap %7.1f  dur %8.1f us  grid %6s  %s' % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r['Grid_Size_X'], r['Kernel_Name'][:50]))
    prev_end = e
