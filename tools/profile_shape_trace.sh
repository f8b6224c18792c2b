#!/bin/bash
# kernel trace of whole reconstructions (tools/time_shapes.py): GPU-busy time against wall time per shape -- what the host leaves idle
#   tools/profile_shape_trace.sh <tag>
TAG=${1:-r4}
OUT=$PWD/gpurun_out/shape_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/tools/time_shapes.py 5"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
cd - > /dev/null
python - "$OUT" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/trace/*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if 'kernel_dispatch' in t]
print('tables:', kd[:4])
t = 'kernels' if 'kernels' in tabs else kd[0]
cols = [r[1] for r in c.execute('pragma table_info({})'.format(t))]
print(t, cols[:20])
PY
python tools/rocpd_summary.py $OUT $OUT/summary > /dev/null
grep "shape" $OUT/trace.log | tail -5
head -12 $OUT/summary_rocprof_summary.txt
find $OUT -name "*.db" -size +60M -delete
