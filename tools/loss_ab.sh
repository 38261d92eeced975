# kernel-trace of tools/loss_bench.py for csrc/libgsr_hip.so (new) and csrc/libgsr_hip_prev.so (prev)
D=3dgs_hierarchical_training_amd/csrc
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp $D/libgsr_hip.so /tmp/new.so; cp $D/libgsr_hip_prev.so /tmp/prev.so
for w in new prev; do cp /tmp/$w.so $D/libgsr_hip.so
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_loss_$w -o r -- python tools/loss_bench.py $@ > /dev/null 2>&1
python - <<PY
import sqlite3
cur=sqlite3.connect("gpurun_out/prof_loss_$w/r_results.db").cursor()
for r in cur.execute("select name,count(*),avg(end-start),min(end-start) from kernels where name like '%k_loss%' group by name"):
    print("$w", r[0][:24], r[1], round(r[2]/1e3,2), round(r[3]/1e3,2))
PY
done
cp /tmp/new.so $D/libgsr_hip.so
