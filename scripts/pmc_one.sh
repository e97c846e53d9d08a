#!/bin/bash
# SQ counters of the kernels whose name contains $1, from one bench.py step (on the GPU box): gpurun -- 'bash scripts/pmc_one.sh map_stream'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SSM_BENCH_H2D=0 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/p_one -o runc -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-other-configs > gpurun_out/p_one.log 2>&1
python3 - "$1" <<'PY'
import csv,glob,collections,sys
pat=sys.argv[1]
f=glob.glob("gpurun_out/p_one/**/*counter_collection.csv",recursive=True)[0]
tr=glob.glob("gpurun_out/p_one/**/*kernel_trace.csv",recursive=True)[0]
dur={r["Dispatch_Id"]:int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))}
agg=collections.defaultdict(lambda: collections.defaultdict(float)); ns=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); ns[k]+=dur.get(r["Dispatch_Id"],0)
for k,c in agg.items():
    if pat not in k: continue
    w=c["SQ_WAVES"]; t=ns[k]*1e-9; clk=c["SQ_BUSY_CYCLES"]/32/t
    print(k[:40], "ms %.3f waves %d VALU/w %.0f LDS/w %.0f SALU/w %.0f cyc/w(x4) %.0f ipc %.3f bankconf/w %.0f waitlds/w %.0f" % (ns[k]/1e6, w, c["SQ_INSTS_VALU"]/w, c["SQ_INSTS_LDS"]/w, c["SQ_INSTS_SALU"]/w, c["SQ_WAVE_CYCLES"]/w, c["SQ_INSTS_VALU"]/(t*clk*1024), c["SQ_LDS_BANK_CONFLICT"]/w, c["SQ_WAIT_INST_LDS"]/w))
PY
