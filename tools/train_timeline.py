"""Phases of one W-HMR training step from a rocprofv3 --kernel-trace database (bench.py --workload whmr_train): ViT forward | heads forward + loss +
heads backward | ViT backward | optimizer, with the busy time of every stream inside each phase.  usage: python tools/train_timeline.py <results.db>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == 'kernels'][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kt)]
key = 'queue_id' if 'queue_id' in cols else 'stream_id'      # rocprofv3 7.x reports stream_id 0 for every launch; the HSA queue tells the streams apart
rows = cur.execute('select start, end, name, %s from %s order by start' % (key, kt)).fetchall()
adam = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r[2] and 'Adam' in r[2]]
# steps end with the last Adam kernel of a group; take the last complete step
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
# the shortest complete step (the last one carries bench.py's instrumented passes)
cands = [(rows[ends[i + 1]][1] - rows[ends[i] + 1][0], ends[i] + 1, ends[i + 1] + 1) for i in range(len(ends) - 1)]
_, lo, hi = min(cands)
step = rows[lo:hi]
t0 = step[0][0]
first = lambda pat: next(r for r in step if pat in r[2])
last = lambda pat: [r for r in step if pat in r[2]][-1]
a = last('attention_bf16_chunk')[1]                 # end of the ViT forward's last attention (+ ~4 launches)
b = first('attention_bwd_bf16')[0]                  # ViT backward's first attention backward (- ~4 launches)
c = next(r for r in step if 'multi_tensor_apply' in r[2] and 'Adam' in r[2])[0]
names = (('ViT forward', t0, a), ('heads fwd + loss + heads bwd', a, b), ('ViT backward', b, c), ('optimizer', c, step[-1][1]))
print('step: %d kernels, %.2f ms' % (len(step), (step[-1][1] - t0) / 1e6))
for nm, s, e in names:
    seg = [r for r in step if r[0] >= s and r[0] < e]
    per = {}
    for r in seg:
        d = per.setdefault(r[3], [0, 0.0, r[0], r[1]])
        d[0] += 1; d[1] += r[1] - r[0]; d[3] = max(d[3], r[1])
    print('%-30s %6.2f ms  %4d kernels' % (nm, (e - s) / 1e6, len(seg)))
    for k, (n, t, f, l) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print('      %s %-4s %4d kernels  busy %.2f ms  active %.2f .. %.2f ms of the phase' % (key, k, n, t / 1e6, (f - s) / 1e6, (l - s) / 1e6))
if len(sys.argv) > 2 and sys.argv[2] == 'names':       # launches of the heads phase by kernel name
    from collections import Counter
    seg = [r for r in step if r[0] >= a and r[0] < b]
    cnt, tim = Counter(), Counter()
    for r in seg:
        cnt[r[2][:110]] += 1; tim[r[2][:110]] += r[1] - r[0]
    for nmk, n in cnt.most_common(70):
        print('%4d  %8.1f us  %s' % (n, tim[nmk] / 1e3, nmk))
elif len(sys.argv) > 2:                              # the heads phase in 0.25 ms buckets: busy time per stream
    s, e = a, b
    seg = [r for r in step if r[0] >= s and r[0] < e]
    streams = sorted({r[3] for r in seg})
    nb = int((e - s) / 250e3) + 1
    for i in range(nb):
        lo_t, hi_t = s + i * 250e3, s + (i + 1) * 250e3
        line = '%5.2f ms ' % (i * 0.25)
        for st in streams:
            bt = sum(min(r[1], hi_t) - max(r[0], lo_t) for r in seg if r[3] == st and r[1] > lo_t and r[0] < hi_t)
            top = max((r for r in seg if r[3] == st and r[1] > lo_t and r[0] < hi_t), key=lambda r: min(r[1], hi_t) - max(r[0], lo_t), default=None)
            line += ' | %s %3.0f%% %-28s' % (st, 100 * bt / 250e3, (top[2][:28] if top else ''))
        print(line)
