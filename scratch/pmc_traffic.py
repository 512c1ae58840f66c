"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) of bench.py into profiles/r01_hbm_traffic_pmc.json and the
per-kernel CSVs.  usage: python scratch/pmc_traffic.py <dir_fetch> <dir_write> <steps_profiled>
FETCH_SIZE / WRITE_SIZE count in units of 32 B? no: rocprofv3 reports them in KB on gfx950 (MI355X_MICROARCH.md HBM section);
calibration against kernels with known traffic is recorded in the json (_calibration)."""
import collections, csv, json, re, sys

FAMILIES = [
    ("conv3x3_fwd_mfma", r"conv3x3_fwd_mfma_v2"), ("conv3x3_wgrad_mfma", r"conv3x3_wgrad_mfma_v2"), ("slab_reduce", r"slab_reduce_kernel"),
    ("bn_relu_apply", r"bn_relu_apply_kernel"), ("bn_relu_bwd_apply", r"bn_relu_bwd_apply_kernel"), ("bn_relu_bwd_reduce", r"bn_relu_bwd_reduce_kernel"),
    ("maxpool2_fwd", r"maxpool2_fwd_kernel"), ("maxpool2_bwd", r"maxpool2_bwd_kernel"), ("upsample2x_fwd", r"upsample2x_fwd_kernel"),
    ("upsample2x_bwd", r"upsample2x_bwd"), ("conv3x3_fwd_stem", r"conv3x3_fwd_stem"), ("conv3x3_wgrad_stem", r"conv3x3_wgrad_stem"),
    ("conv1x1_fwd", r"conv1x1_fwd_kernel"), ("conv1x1_dgrad", r"conv1x1_dgrad_kernel"), ("conv1x1_wgrad", r"conv1x1_wgrad_kernel"),
    ("rmsprop", r"rmsprop_kernel"), ("grad_sumsq", r"grad_sumsq_kernel"), ("pack_w3x3_batched", r"pack_w3x3_batched_kernel"),
    ("bce_dice_sums", r"bce_dice_sums_kernel"), ("boundary_count", r"boundary_count_kernel"),
]


def load(d, counter):
    rows = list(csv.DictReader(open(f"{d}/run_counter_collection.csv")))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        fam = next((f for f, pat in FAMILIES if re.search(pat, r["Kernel_Name"])), None)
        if fam is None:
            continue
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
    return agg


def main():
    dfetch, dwrite, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    f, w = load(dfetch, "FETCH_SIZE"), load(dwrite, "WRITE_SIZE")
    fams = {}
    for fam in sorted(set(f) | set(w)):
        nf, vf = f.get(fam, [0, 0.0]); nw, vw = w.get(fam, [0, 0.0])
        n = max(nf, nw, 1)
        fams[fam] = {"launches_per_step": round(n / steps, 2), "fetch_MB_per_launch_raw": round(vf / max(nf, 1) / 1e3, 2),
                     "fetch_MB_per_launch_x2": round(2 * vf / max(nf, 1) / 1e3, 2), "write_MB_per_launch": round(vw / max(nw, 1) / 1e3, 2)}
    out = {
        "_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 2 --warmup 1 "
                "--no-cpu-baseline --no-kernel-profile; counter values are KB; MB per launch averaged over the launches of a kernel family",
        "_calibration": "WRITE_SIZE is exact. FETCH_SIZE: fully coalesced 16 B/lane streaming kernels (rmsprop: 4 x 69 MB arrays, "
                        "bn_relu_apply) read exactly 2x the raw value, as MI355X_MICROARCH.md says -> use fetch_x2 for them. The conv kernels "
                        "fetch 64-byte segments (4 lanes x 16 B per pixel chunk) by buffer_load...lds: per-layer raw values equal the known "
                        "input bytes -> use fetch_raw for conv3x3_*_mfma.",
        "families": fams,
    }
    for fam in ("conv3x3_fwd_mfma", "conv3x3_wgrad_mfma"):
        if fam in fams:
            out[fam + "_per_launch"] = {"hbm_MB": round(fams[fam]["fetch_MB_per_launch_raw"] + fams[fam]["write_MB_per_launch"], 1)}
    json.dump(out, open("profiles/r01_hbm_traffic_pmc.json", "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
