"""Timeline of one training iteration from a rocprofv3 kernel trace: which kernels ran beside each other.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/train_bench.py
  python3 tools/trace_timeline.py <dir>/.../t_kernel_trace.csv [first kernel index of the window] [count]
Prints start (us, relative), duration, queue and the kernel name; a '|' in front marks a kernel that started before the previous
launch-stream kernel ended (it ran beside it)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
t0 = int(rows[first]['Start_Timestamp'])
last_end = 0
for r in rows[first:first + count]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    mark = '|' if s < last_end else ' '
    print('%s %9.1f us +%8.1f us  q%-3s %s' % (mark, (s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
    last_end = max(last_end, e)
