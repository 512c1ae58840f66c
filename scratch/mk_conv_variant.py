"""Scratch builds of csrc/conv3x3.hip for the round-3 diagnostics (the shipped source is not touched; the variant is written to
/tmp/<name>/, compiled with the library's flags, linted like the shipped kernel and linked with the other objects of csrc/build/
into scratch/ab/lib_<name>.so -- same C ABI, load it with UH_LIB_PATH or scratch/ab_conv.py):
    python scratch/mk_conv_variant.py diag      s_memtime stamps per phase of conv3x3_fwd_mfma_v2 + uh_diag_set(ptr)   (scratch/diag_phases.py)
    python scratch/mk_conv_variant.py diagmin   entry / exit stamps only (in-kernel clock and wave lifetimes of the unperturbed schedule)
    python scratch/mk_conv_variant.py wgdiag    backward-weights: stamps at entry / first fence / end of the tile loop / slab stores drained   (scratch/diag_wgrad.py)
    python scratch/mk_conv_variant.py nwr2      backward-weights with 4-wave workgroups (two per CU) on every layer
    python scratch/mk_conv_variant.py nbw1      forward / backward-data with 16 channels per wave (three workgroups per CU) on every layer
    python scratch/mk_conv_variant.py prio      the workgroup that has finished fewer tiles gets the higher issue priority (s_setprio)
    python scratch/mk_conv_variant.py young     the second half of the grid (dispatched last) gets issue priority
    python scratch/mk_conv_variant.py split31   launches with two tiles per workgroup: first half of the lanes three tiles, second half one"""
import glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd", "csrc")
STAMP = ('#define STAMP_REAL() do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && diag_i < 64) '
         'diag_st[threadIdx.x >> 6][diag_i] = t_; ++diag_i; } while (0)\n')


def rep(s, old, new, count=1):
    assert s.count(old) >= 1, old[:60]
    return s.replace(old, new, count)


def variant(name, s):
    loop = "    for (; tile < ntile; tile += nlanes) {\n        const int next_tile = tile + nlanes;"
    if name in ("diag", "diagmin"):
        s = rep(s, "constexpr int PRE_MAX_C = 512;", "__device__ unsigned long long* uh_diag_buf = nullptr;\n"
                'extern "C" int uh_diag_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(uh_diag_buf), &p, sizeof(p)); }\nconstexpr int PRE_MAX_C = 512;')
        a = "    __shared__ float wg_sum[3][BN];          // [0] = S1, [1] = S2, [2] = pivot; slot = channel - co_blk\n"
        s = rep(s, a, a + "    __shared__ unsigned long long diag_st[4][64];\n    int diag_i = 0;\n" + STAMP +
                ("#define STAMP() STAMP_REAL()\n" if name == "diag" else "#define STAMP() do { } while (0)\n"))
        i = s.index(a)
        nt = "    const int ntile = B * tilesX * tilesY;\n"
        j = s.index(nt, i)
        s = s[:j] + nt + "    const unsigned long long diag_r0 = __builtin_amdgcn_s_memrealtime();\n    STAMP_REAL();\n" + s[j + len(nt):]
        s = rep(s, "    chunk_fence(chunk_of(0), 0, true);\n", "    chunk_fence(chunk_of(0), 0, true);\n    STAMP();\n")
        c = "            mma_shift(buf, 2, wsel(wC));\n            // DMA and WA' landed; every wave has finished reading this buffer\n            chunk_fence(next_c, bufi ^ 1, live_next);\n"
        s = rep(s, c, "            mma_shift(buf, 2, wsel(wC));\n            STAMP();\n            chunk_fence(next_c, bufi ^ 1, live_next);\n            STAMP();\n")
        d = "        // ---- stores, straight from the registers through a buffer descriptor: per-lane byte offset (column, channel\n"
        s = rep(s, d, "        STAMP();\n" + d)
        e = "        // ---- BatchNorm statistics: pivot-shifted sums per lane, 4 DPP adds per channel, accumulated in LDS by the lane that\n"
        s = rep(s, e, "        STAMP();\n" + e)
        f = "            n_run += (float)(vy * vx);\n        }\n"
        s = rep(s, f, f + "        STAMP();\n")
        g = "    if (stats) {\n        // row = tile lane of this workgroup; rows nlanes .. ntile-1"
        s = rep(s, g, ("    STAMP_REAL();\n" if name == "diagmin" else "") +
                "    if (uh_diag_buf) {\n        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;\n"
                "        unsigned long long* o = uh_diag_buf + ((size_t)blockIdx.x * 4 + wv) * 66;\n"
                "        if (ln < diag_i && ln < 64) o[2 + ln] = diag_st[wv][ln];\n"
                "        if (ln == 0) { o[0] = (unsigned long long)diag_i; o[1] = __builtin_amdgcn_s_memrealtime() - diag_r0; }\n    }\n" + g)
    elif name == "wgdiag":
        # backward-weights: entry, behind the first fence, behind the tile loop, behind the slab stores (per wave)
        s = rep(s, "constexpr int PRE_MAX_C = 512;", "__device__ unsigned long long* uh_diag_buf = nullptr;\n"
                'extern "C" int uh_diag_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(uh_diag_buf), &p, sizeof(p)); }\nconstexpr int PRE_MAX_C = 512;')
        k = s.index("void conv3x3_wgrad_mfma_v2(")
        a = "    const int tid = threadIdx.x, lane = tid & 63;\n"
        j = s.index(a, k)
        s = s[:j] + a + "    unsigned long long dg[4]; const unsigned long long dg_r0 = __builtin_amdgcn_s_memrealtime(); dg[0] = __builtin_amdgcn_s_memtime();\n" + s[j + len(a):]
        b = "    tile_fence(t_begin, 0, t_begin < t_end);\n    int bufi = 0;\n"
        j = s.index(b, k)
        s = s[:j] + b + "    dg[1] = __builtin_amdgcn_s_memtime();\n" + s[j + len(b):]
        f = "        tile_fence(tile + 1, bufi ^ 1, tile + 1 < t_end);\n    }\n"
        j = s.index(f, k)
        s = s[:j] + "        { const unsigned long long f0 = __builtin_amdgcn_s_memtime();\n        tile_fence(tile + 1, bufi ^ 1, tile + 1 < t_end);\n" \
            "        dg_fence += __builtin_amdgcn_s_memtime() - f0; }\n    }\n" + s[j + len(f):]
        s = s.replace("unsigned long long dg[4]; const unsigned long long dg_r0", "unsigned long long dg[4]; unsigned long long dg_fence = 0; const unsigned long long dg_r0", 1)
        c = "    float* slab = slabs + (int64_t)split * Cout * 9 * Cin;\n#if UH_WGRAD_M16\n"
        j = s.index(c, k)
        s = s[:j] + "    dg[2] = __builtin_amdgcn_s_memtime();\n" + c + s[j + len(c):]
        # end of kernel: the closing of the #else / #endif slab store
        d = "            slab[((int64_t)co * 9 + tap) * Cin + ci] = acc[tap][reg];\n        }\n#endif\n}"
        j = s.index(d, k)
        s = s[:j] + "            slab[((int64_t)co * 9 + tap) * Cin + ci] = acc[tap][reg];\n        }\n#endif\n" \
            "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    dg[3] = __builtin_amdgcn_s_memtime();\n" \
            "    if (uh_diag_buf && lane == 0) {\n        unsigned long long* o = uh_diag_buf + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + (tid >> 6)) * 8;\n" \
            "        o[0] = 4; o[1] = __builtin_amdgcn_s_memrealtime() - dg_r0; o[2] = dg[0]; o[3] = dg[1]; o[4] = dg[2]; o[5] = dg[3]; o[6] = (unsigned long long)(t_end - t_begin); o[7] = dg_fence;\n    }\n}" + s[j + len(d):]
    elif name == "nwr2":
        # backward-weights: 4-wave workgroups (64 x 64 channel tiles, two per CU) on every layer
        s = rep(s, "        if (wide && sizeof(T) == 2 && Cout % 128 == 0 && !half_cu) p.nwr = 4;", "        (void)wide;")
    elif name == "nbw1":
        # forward / backward-data: the 64-channel-slab instantiation (16 channels per wave, 168 registers, THREE workgroups per CU) on every layer
        s = s.replace("if (Cout % 128 == 0 && (int64_t)ntile * (Cout / 128) >= 512) {", "if (false) {")
        s = s.replace("    if (Cout % 128 == 0 && (int64_t)ntile * (Cout / 128) >= 512) { r.nbw = 2;", "    if (false) { r.nbw = 2;")
    elif name == "prio":
        s = rep(s, loop, "    int tiles_done = 0;\n" + loop + "\n        if (tiles_done == 0) __builtin_amdgcn_s_setprio(3);\n"
                "        else if (tiles_done == 1) __builtin_amdgcn_s_setprio(2);\n        else if (tiles_done == 2) __builtin_amdgcn_s_setprio(1);\n"
                "        else __builtin_amdgcn_s_setprio(0);\n        ++tiles_done;")
    elif name == "young":
        s = rep(s, loop, "    if (tile_lane >= (nlanes >> 1)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);\n" + loop)
    elif name == "split31":
        s = rep(s, "    int tile = tile_lane;\n    if (tile >= ntile) return;\n",
                "    int tile = tile_lane;\n    if (tile >= ntile) return;\n    int tstride = nlanes, tend = ntile;\n"
                "    if (ntile == 2 * nlanes && (nlanes & 1) == 0) {\n        const int hh = nlanes >> 1;\n"
                "        if (tile_lane < hh) { tstride = hh; tend = 3 * hh; } else { tile = 2 * hh + tile_lane; }\n    }\n")
        s = rep(s, loop, "    for (; tile < tend; tile += tstride) {\n        const int next_tile = tile + tstride;")
        s = s.replace("live_next = next_tile < ntile;", "live_next = next_tile < tend;")
    else:
        raise SystemExit(__doc__)
    return s


def main():
    name = sys.argv[1]
    d = f"/tmp/{name}"
    os.makedirs(d, exist_ok=True)
    src = variant(name, open(os.path.join(CSRC, "conv3x3.hip")).read())
    open(f"{d}/{name}.hip", "w").write(src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-Wno-inline-asm", "-Wno-unused-value", "-Wno-unused-result", "-save-temps=obj", "-c", f"{d}/{name}.hip", "-o", f"{d}/{name}.o"]
    subprocess.run(cmd, check=True, cwd=d, stderr=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd"))
    import isa_lint
    errs, _ = isa_lint.lint_asm(open(f"{d}/{name}-hip-amdgcn-amd-amdhsa-gfx950.s").read())
    print("isa lint:", errs or "clean")
    objs = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o")) if not o.endswith("conv3x3.o")]
    os.makedirs(os.path.join(ROOT, "scratch", "ab"), exist_ok=True)
    out = os.path.join(ROOT, "scratch", "ab", f"lib_{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + [f"{d}/{name}.o"], check=True)
    print(out)


if __name__ == "__main__":
    main()
