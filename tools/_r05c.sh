O=gpurun_out/r05c; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/$O/trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 28 --warmup 14 --no-cpu-baseline --deliver > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/trace.log | cut -c1-300
find $O/trace -name "*.csv" | head; 
python3 - <<'PY'
import csv,glob
k=glob.glob('gpurun_out/r05c/trace/**/*kernel_trace.csv',recursive=True)[0]
m=glob.glob('gpurun_out/r05c/trace/**/*memory_copy_trace.csv',recursive=True)[0]
ev=[]
rows=list(csv.DictReader(open(k)))
print(rows[0].keys())
for r in rows:
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('dabx::','')[:28], r.get('Queue_Id','')))
mrows=list(csv.DictReader(open(m)))
print(mrows[0].keys() if mrows else 'no memcpy rows')
for r in mrows:
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'MEMCPY '+r.get('Direction','')+' '+str(r.get('Bytes',r.get('Size',''))), ''))
ev.sort()
t_end=ev[-1][1]
sel=[e for e in ev if e[0]>t_end-60e6 and (e[2].startswith('MEMCPY') or 'deliver' in e[2] or 'vitT' in e[2] or 'dabplus' in e[2] or 'frame_tail' in e[2])]
t0=sel[0][0]
with open('gpurun_out/r05c/timeline.txt','w') as f:
    for e in sel: f.write("%9.3f %9.3f %8.3f %s q%s\n"%((e[0]-t0)/1e6,(e[1]-t0)/1e6,(e[1]-e[0])/1e6,e[2],e[3]))
PY
tail -150 $O/timeline.txt
