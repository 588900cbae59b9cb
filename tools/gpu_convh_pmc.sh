cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmch; export TMPDIR=/tmp
R=$PWD
for C in 64 128; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmch/a$C -o p -- python tools/pair_bench.py $C 0 1 split > gpurun_out/pmch/a$C.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmch/b$C -o p -- python tools/pair_bench.py $C 0 1 split > gpurun_out/pmch/b$C.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmch/*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]+" g"+r.get("Grid_Size","")+" lds"+r.get("LDS_Block_Size","") ; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    print("==",f)
    for k,d in agg.items():
        if "convh" not in k: continue
        print(k)
        for c,v in sorted(d.items()): print(f"   {c:28s} {v/cnt[(k,c)]:16.0f} (n={cnt[(k,c)]})")
PY
