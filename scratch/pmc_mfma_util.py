"""rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE --kernel-trace output of scratch/kprof2.py (bench_double_conv, iters=3) -> one CSV row
per dispatch of the six conv kernels of the 256-channel DoubleConv.   usage: python scratch/pmc_mfma_util.py DIR > out.csv"""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = rows.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]),
                                           "vgpr": r.get("VGPR_Count"), "lds": r.get("LDS_Block_Size"), "c": {}})
    e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    e["dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
B, H, W, Ci, Co = 8, 128, 128, 128, 256
f1, f2 = 2.0 * B * H * W * Co * 9 * Ci, 2.0 * B * H * W * Co * 9 * Co
labels = ([("fwd_conv1 128->256", f1)] * 4 + [("fwd_conv2 256->256", f2)] * 4 + [("dgrad_conv2 256->256", f2)] * 4 + [("dgrad_conv1 256->128", f1)] * 4 +
          [("wgrad_conv2 256x256", f2)] * 4 + [("wgrad_conv1 256x128", f1)] * 4)
conv = [e for e in rows.values() if "conv3x3_fwd_mfma_v2" in e["name"] or "conv3x3_wgrad_mfma_v2" in e["name"]]
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -- python scratch/kprof2.py")
print("# 256-channel DoubleConv (down2: 128->256->256 @128x128, batch 8, bf16), one row per dispatch (1 warm-up + 3 timed per kernel).")
print("# clock_gui_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration (reads high on dispatches this short: MI355X_MICROARCH.md); mfma_util_lo = MFMA busy cycles per SIMD /")
print("# (clock_gui * duration) is therefore a LOWER bound of the matrix-pipe utilisation; mfma_ghz_equiv = busy cycles per SIMD / duration = the clock at which")
print("# the pipe would have to run to do this work with no idle cycle (util x clock), the quantity the TFLOP/s follow.")
print("layer,kernel,grid,wg,vgpr,lds_bytes,dur_us,tflops,frac_of_2500,clock_gui_ghz,mfma_busy_cycles_per_simd,mfma_util_lo,mfma_ghz_equiv,wait_any_frac,wait_inst_any_frac,active_inst_frac")
for (lab, fl), e in zip(labels, conv[:len(labels)]):
    c = e["c"]; dur = e["dur"]
    clk = c.get("GRBM_GUI_ACTIVE", float("nan")) / 8 / (dur * 1e3)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / 1024
    wc = c.get("SQ_WAVE_CYCLES", float("nan"))
    kn = "conv3x3_wgrad_mfma_v2" if "wgrad" in e["name"] else "conv3x3_fwd_mfma_v2"
    print(f"{lab},{kn},{e['grid']},{e['wg']},{e['vgpr']},{e['lds']},{dur:.1f},{fl / dur / 1e6:.0f},{fl / dur / 1e6 / 2500:.3f},{clk:.2f},{busy:.0f},"
          f"{busy / (clk * dur * 1e3):.3f},{busy / (dur * 1e3):.3f},{c.get('SQ_WAIT_ANY', 0) / wc:.3f},{c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f},{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}")
