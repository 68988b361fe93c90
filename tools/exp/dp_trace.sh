cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613 DENET_FORCE_DP=1 rocprofv3 --kernel-trace -d gpurun_out/r02_dp -o kt -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-warm --no-roofline > gpurun_out/r02_dp.log 2>&1
DB=$(find gpurun_out/r02_dp -name "*.db" | head -1)
python tools/timeline.py $DB 6
python - <<PY
import sqlite3
db=sqlite3.connect("$DB")
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T=lambda p:[t for t in tabs if t.startswith(p)][0]
rows=db.execute("select s.kernel_name, d.queue_id, d.stream_id, count(*), sum(d.end-d.start)/1e6 from %s d join %s s on d.kernel_id=s.id group by 1,2,3 order by 5 desc"%(T("rocpd_kernel_dispatch"),T("rocpd_info_kernel_symbol"))).fetchall()
for r in rows[:14]: print(r[0][-50:], r[1:])
print("queues/streams:", db.execute("select queue_id, stream_id, count(*), sum(end-start)/1e6 from %s group by 1,2"%T("rocpd_kernel_dispatch")).fetchall())
PY
rm -rf gpurun_out/r02_dp
