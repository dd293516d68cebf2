# SQ counters of the large search on maps whose records do not fit LDS (the gather6 form):
# 1 M candidates x 720 beams on the cfg-3 and cfg-5 maps (experiments/big_map_search.py).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bigmap
rm -rf $O && mkdir -p $O
python3 $R/experiments/big_map_search.py > $O/plain.json 2> $O/plain.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY --output-format csv -d $O/sq1 -- python3 $R/experiments/big_map_search.py > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $O/sq2 -- python3 $R/experiments/big_map_search.py > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_bigmap"
for d in ("sq1","sq2"):
    for path in glob.glob(O+"/"+d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k=r["Kernel_Name"]
            if "match_lane" in k and "combine" not in k:
                key=k.split("::")[-1].split("(")[0]
                agg[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), float(r["End_Timestamp"])-float(r["Start_Timestamp"]), r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size"]))
        for k,c in agg.items():
            for name,vals in sorted(c.items()):
                print(d,k,name," ".join("%.4g(%.0fus)"%(v[1],v[2]/1e3) for v in vals[-4:]), "vgpr",vals[-1][3],"scratch",vals[-1][4],"lds",vals[-1][5],"grid",vals[-1][6])
PY
cat $O/plain.json | cut -c1-600
