"""Print the kernel timeline of one graph-replayed frame from a rocprofv3 kernel trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'(void )?(\w+)(<[^>]*>)?', n)
    return (m.group(2) + (m.group(3) or '')) if m else n[:40]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [short(r['Kernel_Name']) for r in rows]
starts = [i for i, n in enumerate(names) if n.startswith('vox_fill')]
i0, i1 = starts[which], starts[which + 1]
t0 = int(rows[i0]['Start_Timestamp'])
print('frames', len(starts), 'frame span us', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
for r in rows[i0:i1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    print('%8.1f %8.1f %7.1f  q%s %s' % (s, e, e - s, r.get('Queue_Id', '?'), short(r['Kernel_Name'])[:70]))
