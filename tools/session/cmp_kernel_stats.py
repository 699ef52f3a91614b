import csv,glob,os,re,sys
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['Name']]=(int(r['Calls']),float(r['AverageNs']),float(r['TotalDurationNs']))
    return d
old=load(max(glob.glob('/root/repo/gpurun_out/r3m_old_cfg2/*/*kernel_stats.csv'),key=os.path.getmtime))
new=load(max(glob.glob('/root/repo/gpurun_out/r3m_new_cfg2/*/*kernel_stats.csv'),key=os.path.getmtime))
def norm(k):
    k=k.replace('conv16_ws_kernel','conv16_kernel')
    k=re.sub(r'(Conv16Cfg<(?:\d+, ){8}\d+), [01]>',r'\1>',k)
    k=k.replace('KparCfg<3, 3>','KparCfg<3, 3, 1, 0>').replace('KparCfg<1, 3>','KparCfg<1, 3, 1, 0>')
    return k.split('(')[0] if ('gn_finalize2' in k or 'linear_kernel' in k or 'update_kernel' in k) else k
o={};n={}
for k,v in old.items(): o[norm(k)]=v
for k,v in new.items(): n[norm(k)]=v
print("per-forward total old/new", sum(v[2] for v in o.values())/1e6/12, sum(v[2] for v in n.values())/1e6/12)
for k in sorted(set(o)|set(n), key=lambda k:-max(o.get(k,(0,0,0))[2],n.get(k,(0,0,0))[2]))[:int(sys.argv[1]) if len(sys.argv)>1 else 16]:
    a=o.get(k,(0,0,0)); b=n.get(k,(0,0,0))
    print(f"{k[:92]:92s} old {a[0]:4d} x {a[1]/1e3:7.1f}   new {b[0]:4d} x {b[1]/1e3:7.1f}  d/fwd {(b[2]-a[2])/1e6/12:+.3f} ms")
