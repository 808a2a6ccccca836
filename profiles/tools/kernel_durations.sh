#!/bin/bash
# Distribution of one kernel's dispatch durations over a bench run (rocprofv3
# kernel trace, CSV): kernel_durations.sh PATTERN bench args...
pat=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kd
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kd -o out -- \
    python $R/bench.py "$@" --no-cpu-baseline --no-secondary > /dev/null 2> /tmp/kd.err
python - "$pat" <<'PY'
import csv, glob, re, sys
pat = sys.argv[1]
f = glob.glob('/tmp/kd/**/*kernel_trace.csv', recursive=True)[0]
d = []
for r in csv.DictReader(open(f)):
    if re.search(pat, r['Kernel_Name']):
        d.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
d2 = sorted(d)
n = len(d2)
print(pat, 'dispatches', n)
if n:
    for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0):
        print(' q%.2f %.1f us' % (q, d2[min(n - 1, int(q * n))]))
    print(' last 40 in order:', ' '.join('%.0f' % x for x in d[-40:]))
PY
