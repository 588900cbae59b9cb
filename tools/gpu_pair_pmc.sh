cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmcp; export TMPDIR=/tmp
R=$PWD
for C in 16 32; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmcp/a$C -o p -- python tools/pair_bench.py $C > gpurun_out/pmcp/a$C.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcp/b$C -o p -- python tools/pair_bench.py $C > gpurun_out/pmcp/b$C.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcp/st -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/pmcp/st.log 2>&1
ls -R gpurun_out/pmcp | head -40
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcp/*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    print("==",f)
    for k,d in agg.items():
        if "pair" not in k: continue
        print(k)
        for c,v in sorted(d.items()): print(f"   {c:28s} {v/cnt[(k,c)]:16.0f} (n={cnt[(k,c)]})")
PY
