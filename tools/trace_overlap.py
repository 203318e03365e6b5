"""Start / end of every dispatch in a rocprofv3 kernel trace, relative to the first: python tools/trace_overlap.py <dir>"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -16:]
for r in last:
    m = re.search(r'(\w+_kernel\w*)', r['Kernel_Name'])
    print('%-28s start %9.1f us  end %9.1f us  dur %7.1f us  queue %s' % (m.group(1) if m else r['Kernel_Name'][:28],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?')))
