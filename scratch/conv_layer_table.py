"""profiles/rNN_conv_layer_table.md from a kernel trace of the train step (scratch/profile_round.sh stats pass, exported with
`scratch/rocpd_export.py trace`): the 51 3x3 MFMA conv launches of one step of UNet(1,1,bilinear) B=8 1x512x512 bf16, by layer.
    python scratch/conv_layer_table.py <trace.csv> <steps in the trace> <out.md> "<bench line note>"
Launch order inside a step: 17 forward launches in layer order, then backward in reverse layer order (backward-data and
backward-weights families each keep that order)."""
import csv, re, sys

LAYERS = [("inc.3", 512, 64, 64), ("down1.0", 256, 64, 128), ("down1.3", 256, 128, 128), ("down2.0", 128, 128, 256), ("down2.3", 128, 256, 256),
          ("down3.0", 64, 256, 512), ("down3.3", 64, 512, 512), ("down4.0", 32, 512, 512), ("down4.3", 32, 512, 512),
          ("up1.0", 64, 1024, 512), ("up1.3", 64, 512, 256), ("up2.0", 128, 512, 256), ("up2.3", 128, 256, 128),
          ("up3.0", 256, 256, 128), ("up3.3", 256, 128, 64), ("up4.0", 512, 128, 64), ("up4.3", 512, 64, 64)]
B = 8                                  # sixth argument overrides (the batch-4 leg)


def is_bnsum(name: str) -> bool:
    """the BSUM template argument of conv3x3_fwd_mfma_v2 (second to last since round 4: <T, NBW, SPLIT, WRES, PRE, BSUM, KS>),
    mangled or demangled spelling"""
    m = re.search(r"ELb([01])ELi[12]EEvPKT", name) or re.search(r"ELb([01])EEvPKT", name)
    if m:
        return m.group(1) == "1"
    m = re.search(r"conv3x3_fwd_mfma_v2<([^>]*)>", name)
    if not m:
        return False
    args = [a.strip() for a in m.group(1).split(",")]
    return (args[-2] if args[-1] in ("1", "2") else args[-1]) == "true"


def main():
    global B
    trace, nsteps, out, note = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if len(sys.argv) > 5:
        B = int(sys.argv[5])
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    fam = [r for r in rows if "conv3x3_fwd_mfma_v2" in r["Kernel_Name"]]
    wg = [r for r in rows if "conv3x3_wgrad_mfma_v2" in r["Kernel_Name"]]
    assert len(fam) == 34 * nsteps and len(wg) == 17 * nsteps, (len(fam), len(wg))
    t = {}
    fused = set()
    for s in range(1, nsteps):                      # the first step of the trace is a warm-up step
        f, w = fam[s * 34:(s + 1) * 34], wg[s * 17:(s + 1) * 17]
        for i, l in enumerate(LAYERS):
            t.setdefault(("fwd", l[0]), []).append(int(f[i]["DurationNs"]) / 1e3)
        for i, l in enumerate(reversed(LAYERS)):
            t.setdefault(("dgrad", l[0]), []).append(int(f[17 + i]["DurationNs"]) / 1e3)
            t.setdefault(("wgrad", l[0]), []).append(int(w[i]["DurationNs"]) / 1e3)
            if is_bnsum(f[17 + i]["Kernel_Name"]):
                fused.add(l[0])
    avg = {k: sum(v) / len(v) for k, v in t.items()}
    lines = ["# 3x3 conv launches of one train step, layer by layer", "", note, "",
             "Backward-weights time is the MFMA kernel alone (its `slab_reduce` launch, ~10 us: the slabs are 16-bit pairs, is not included). TFLOP/s = 2*B*H*W*Cout*9*Cin / time. "
             "`*` = backward-data launches that also form the BatchNorm-backward sums of the layer in front (`uh_conv3x3_dgrad_bnsum`): their time includes "
             "what `uh_bn_relu_bwd_reduce` used to spend in a launch of its own.", "",
             "| layer | H=W | Cin | Cout | GFLOP | forward us | TFLOP/s | backward-data us | TFLOP/s | backward-weights us | TFLOP/s |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    gf_tot = 0.0
    for name, hw, ci, co in LAYERS:
        gf = 2.0 * B * hw * hw * co * 9 * ci / 1e9
        gf_tot += gf
        cells = []
        for k in ("fwd", "dgrad", "wgrad"):
            us = avg[(k, name)]
            tot[k] += us
            star = "*" if (k == "dgrad" and name in fused) else ""
            cells += [f"{us:.1f}{star}", f"{gf / us * 1e3:.0f}"]
        lines.append(f"| {name} | {hw} | {ci} | {co} | {gf:.1f} | " + " | ".join(cells) + " |")
    lines.append(f"| **sum** | | | | {gf_tot:.1f} | {tot['fwd']:.0f} | {gf_tot / tot['fwd'] * 1e3:.0f} | {tot['dgrad']:.0f} | {gf_tot / tot['dgrad'] * 1e3:.0f} | "
                 f"{tot['wgrad']:.0f} | {gf_tot / tot['wgrad'] * 1e3:.0f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
