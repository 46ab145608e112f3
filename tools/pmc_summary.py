import sqlite3, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob(sys.argv[1] + '/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    for name, ctr, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
        k = name.split('(')[0].replace('alignnet::', '').replace('void ', '')
        if any(x in k for x in sys.argv[2].split(',')): acc[k][ctr].append(val)
for k, v in sorted(acc.items()):
    print(k)
    for ctr, vals in sorted(v.items()):
        vals = vals[-9:]
        print('   %-32s last3 %s' % (ctr, ['%.4g' % x for x in vals[-3:]]))
