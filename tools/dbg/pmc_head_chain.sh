OUT=/root/repo/gpurun_out/r5i; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python /root/repo/tools/time_head_chain.py"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
python - <<'P'
import csv,glob,collections
for d in ('p1','p2','p3'):
    acc=collections.defaultdict(lambda:[0,0.0])
    for f in glob.glob('/root/repo/gpurun_out/r5i/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'head_chain_fwd' in r['Kernel_Name']:
                a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for k,(n,v) in sorted(acc.items()): print(d,k,n,'avg %.4g'%(v/n))
P
find /root/repo/gpurun_out/r5i -name "*.csv" -delete
