#!/bin/bash
# Which unit the forward conv kernel waits for, shipped library against the no-filter-loads diagnostic build (scratch/libs/libunet_hip_abl_NOW.so):
# two --pmc passes each over scratch/r5_pmc_probe.py.      scratch/r5_pmc_ablate.sh <outdir under gpurun_out> [variant ...]
# (variants = suffixes of scratch/libs/libunet_hip_<variant>.so; default "abl_NOW"; "base" = the shipped library, always first)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
VARS="base ${@:-abl_NOW}"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY"
P2="TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
for v in $VARS; do
  if [ $v = base ]; then unset UH_LIB_PATH; else export UH_LIB_PATH=$R/scratch/libs/libunet_hip_$v.so; fi
  for p in 1 2; do
    if [ $p = 1 ]; then P=$P1; else P=$P2; fi
    rocprofv3 --kernel-trace --pmc $P -d $OUT/${v}_p$p -- python3 $R/scratch/r5_pmc_probe.py > $OUT/${v}_p$p.log 2>&1 || { echo "$v pass $p failed"; tail -5 $OUT/${v}_p$p.log; exit 1; }
    db=$(find $OUT/${v}_p$p -name "*.db" | head -1); mkdir -p $OUT/${v}_c$p; python3 $R/scratch/rocpd_export.py counters $db $OUT/${v}_c$p; rm -rf $OUT/${v}_p$p
    echo "$v pass $p done"
  done
done
unset UH_LIB_PATH
VARS="$VARS" python3 - <<PY
import csv, collections, os
names = ["up1.0 1024->512 @64", "down2.3 256->256 @128", "up4.0 128->64 @512", "inc.3 64->64 @512"]
def load(d):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(d + "/export_counter_collection.csv")):
        if "conv3x3_fwd_mfma_v2" not in r["Kernel_Name"]: continue
        e = rows.setdefault(r["Dispatch_Id"], {"c": {}})
        e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
    return list(rows.values())
out = ["# rocprofv3 --pmc passes over scratch/r5_pmc_probe.py (4 forward conv shapes of config 2, batch 8, bf16; last 3 of 4 launches averaged):",
       "# base = the shipped library; the others = scratch/libs/libunet_hip_<lib>.so (abl_NOW: no filter-fragment loads in the chunk loop, results garbage by design; wearly / strel: scratch/r5/conv_w_early_store_relax.diff).",
       "# mfma_ghz_equiv = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / duration (utilisation x clock); wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; vmem_rd = SQ_INSTS_VMEM_RD per launch;",
       "# ta_busy = TA_BUSY_avr / (GRBM-free estimate: duration x 2.0 GHz) is NOT normalised here: the raw average busy cycles per TA are listed; tcc_hit = TCC_HIT / (HIT + MISS).",
       "layer,lib,dur_us,mfma_ghz_equiv,wait_inst_frac,wait_any_frac,vmem_rd_insts,ta_busy_avr_cycles,tcp_tcc_read_req,tcp_pending_stall_cycles,tcc_hit_rate"]
for v in os.environ["VARS"].split():
    a, b = load("$OUT/%s_c1" % v), load("$OUT/%s_c2" % v)
    for i, n in enumerate(names):
        ga, gb = a[4 * i + 1:4 * i + 4], b[4 * i + 1:4 * i + 4]
        m = lambda g, k: sum(e["c"].get(k, 0.0) for e in g) / len(g)
        dur = sum(e["dur"] for e in ga) / len(ga)
        wc = m(ga, "SQ_WAVE_CYCLES")
        hit, miss = m(gb, "TCC_HIT_sum"), m(gb, "TCC_MISS_sum")
        out.append(f"{n},{v},{dur:.1f},{m(ga, 'SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (dur * 1e3):.3f},{m(ga, 'SQ_WAIT_INST_ANY') / wc:.3f},{m(ga, 'SQ_WAIT_ANY') / wc:.3f},"
                   f"{m(ga, 'SQ_INSTS_VMEM_RD'):.0f},{m(gb, 'TA_BUSY_avr'):.0f},{m(gb, 'TCP_TCC_READ_REQ_sum'):.0f},{m(gb, 'TCP_PENDING_STALL_CYCLES_sum'):.0f},{hit / max(hit + miss, 1):.3f}")
open("$OUT/pmc_ablation.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
