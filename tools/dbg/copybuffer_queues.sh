# which hardware queue the ~90 __amd_rocclr_copyBuffer launches per fit step run on (the step's or the loader's), and their grid sizes
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/cpq; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python $R/tools/time_fit_graph.py --steps 10 > $O/trace.log 2>&1)
python tools/fit_timeline.py $O/trace 10 --per-queue > $O/timeline.txt 2>&1
python tools/fit_timeline.py $O/trace 10 --by-grid > $O/timeline_grid.txt 2>&1
grep -n "copyBuffer" $O/timeline_grid.txt | head -60
python - <<PY
import csv, glob
f = glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step: between the last two adamw launches: print the kernel before and after each copyBuffer on the step queue
marks = [i for i, r in enumerate(rows) if 'adamw_pieces' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
q = rows[b]['Queue_Id']
seq = [r for r in rows[a:b + 1] if r['Queue_Id'] == q]
out = open('$O/copy_neighbours.txt', 'w')
for i, r in enumerate(seq):
    if 'copyBuffer' in r['Kernel_Name']:
        prev = seq[i - 1]['Kernel_Name'][:60] if i else '-'
        nxt = seq[i + 1]['Kernel_Name'][:60] if i + 1 < len(seq) else '-'
        out.write('{:>8} grid {:>8}  {:6.1f} us   after {}   before {}\n'.format(i, r.get('Grid_Size_X', r.get('Grid_Size', '?')), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, prev, nxt))
out.close()
PY
rm -rf $O/trace
