"""Compare the per-parameter gradients written by r5_slab_structured.py: every variant against the fp32-slab run with the same tile
order (the only difference is then the rounding of the partial sums); fp32-contig against fp32-strided gives the floor that
the summation order alone moves.  Yardstick: 2^-9 / sqrt(3) = 1.1e-3, the rms relative error of rounding every element of the
TOTAL to bf16 once -- what the reference's autocast backward does (train.py:116)."""
import torch

d = "/tmp/r5slab"
tags = ("s16_strided", "s16_contig", "f32_strided", "f32_contig")
for which in ("init", "trained"):
    G = {t: torch.load(f"{d}/grads_{which}_{t}.pt") for t in tags}
    print(f"== {which} weights: relative L2 of the 3x3 filter gradients; columns: s16_strided-f32_strided, s16_contig-f32_contig, "
          "f32_contig-f32_strided, bf16(f32_strided)-f32_strided")
    for k in G["f32_strided"]:
        ref = G["f32_strided"][k].double()
        if ref.dim() != 4 or ref.shape[-1] != 3:
            continue
        n = ref.norm().clamp_min(1e-300)
        a = float((G["s16_strided"][k].double() - ref).norm() / n)
        b = float((G["s16_contig"][k].double() - G["f32_contig"][k].double()).norm() / n)
        c = float((G["f32_contig"][k].double() - ref).norm() / n)
        r = float((G["f32_strided"][k].bfloat16().double() - ref).norm() / n)
        print(f"  {k:45s} {a:.2e} {b:.2e} {c:.2e} {r:.2e}")
