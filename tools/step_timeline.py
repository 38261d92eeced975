import sqlite3, sys
cur=sqlite3.connect(sys.argv[1]).cursor()
rows=cur.execute("select name,start,end from kernels order by start").fetchall()
# a step ends with the per-Gaussian backward (+ Adam + next preprocess); the next one starts right behind it
idx=[i+1 for i,r in enumerate(rows) if 'k_preprocess_bwd' in r[0] and i+1 < len(rows)]
k=int(sys.argv[2]) if len(sys.argv)>2 else -4
a,b=idx[k],idx[k+1]
t0=rows[a][1]
prev_end=None; gaps=0
for r in rows[a:b]:
    gap=(r[1]-prev_end)/1e3 if prev_end else 0
    gaps+=max(gap,0)
    print(f"{(r[1]-t0)/1e3:9.1f} +{(r[2]-r[1])/1e3:7.1f}  gap {gap:6.1f}  {r[0][:70]}")
    prev_end=r[2]
print("step span", (rows[b][1]-t0)/1e3, "gaps", gaps + (rows[b][1]-prev_end)/1e3)
