#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd sqlite .db, as written by `rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into a short text file under profiles/.  usage: summarize.py results.db out.txt "command line that was profiled" """
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
with open(sys.argv[2], 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n')
    f.write('# command: %s\n' % (sys.argv[3] if len(sys.argv) > 3 else '?'))
    f.write('%-90s %6s %14s %12s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for i, (name, calls, tot, avg, pct) in enumerate(rows):
        if i >= 12 and 'xp::' not in name:                 # the top twelve, and every kernel of the library
            continue
        name = name if len(name) <= 90 else name[:87] + '...'
        f.write('%-90s %6d %14.1f %12.1f %7.2f\n' % (name, calls, tot, avg, pct))
print(open(sys.argv[2]).read())
