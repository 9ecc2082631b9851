import csv, collections, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r['Kernel_Name'][:46]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[name]['dur'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
ctrs = sorted({r['Counter_Name'] for r in rows})
print('%-46s %8s ' % ('kernel', 'dur_us') + ' '.join('%14s' % c[-14:] for c in ctrs))
for k, v in sorted(agg.items(), key=lambda kv: -st.mean(kv[1]['dur'])):
    if 'at::' in k or 'rocclr' in k: continue
    print('%-46s %8.1f ' % (k, st.mean(v['dur'])) + ' '.join('%14.0f' % st.mean(v[c]) for c in ctrs))
