import sqlite3, glob, sys, collections
for d in sys.argv[1:]:
    c = sqlite3.connect(glob.glob(d + '/**/*.db', recursive=True)[0])
    per = collections.defaultdict(list)
    for n, dur in c.execute("select name,(end-start) from kernels order by start"):
        per[n.split('(')[0].replace('alignnet::','').replace('void ','')].append(dur/1e3)
    out = []
    for k in ('train_fwd_phase23<3>', 'train_bwd_b2', 'train_bwd_b1', 'train_fwd_phase23<2>', 'pointnet_fused'):
        if k in per:
            v = per[k][-9:]
            out.append("%s %s" % (k.replace('train_',''), [round(x) for x in v[-3:]]))
    print(d.split('/')[-1], ' | '.join(out))
