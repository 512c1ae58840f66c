"""Build scratch/ab/libunet_hip_trace.so: conv3x3_fwd_mfma_v2 with wall-clock stamps (100 MHz s_memrealtime) at its phase
boundaries for a few workgroups, read back through uh_trace_read().  The repo sources stay untouched: the patch is applied
to a copy.  Usage: python scratch/trace_patch.py && UH_LIB_PATH=scratch/ab/libunet_hip_trace.so python scratch/trace_run.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unet-medical-image-contour-segmentation_amd")
src = open(os.path.join(PKG, "csrc", "conv3x3.hip")).read()

def rep(old, new, count=1):
    global src
    assert src.count(old) >= 1, old[:60]
    src = src.replace(old, new, count)

rep("constexpr int HALO2_BYTES = HALO_PIX * 64;",
    '''__device__ unsigned long long uh_trace_buf[16 * 128];
extern "C" int uh_trace_read(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(uh_trace_buf), sizeof(unsigned long long) * 16 * 128); }
#define UH_TSTAMP(slot) do { if (tr_on && tr_n < 127) { uh_trace_buf[tr_base + tr_n] = ((unsigned long long)(slot) << 56) | (wall_clock64() & 0x00FFFFFFFFFFFFFFull); ++tr_n; } } while (0)
constexpr int HALO2_BYTES = HALO_PIX * 64;''')
rep("    int bufi = 0;\n\n    for (; tile < ntile; tile += nlanes) {",
    "    int bufi = 0;\n    const bool tr_on = (tid == 0) && (blockIdx.x % 53 == 0) && (blockIdx.x / 53 < 16);\n    const int tr_base = (blockIdx.x / 53) * 128;\n    int tr_n = 0;\n\n    for (; tile < ntile; tile += nlanes) {\n        UH_TSTAMP(1);")
rep("            const unsigned char* buf = lds + bufi * HALO2_BYTES;\n            const int64_t wcp = (int64_t)c * CK;",
    "            UH_TSTAMP(2);\n            const unsigned char* buf = lds + bufi * HALO2_BYTES;\n            const int64_t wcp = (int64_t)c * CK;")
rep("            __syncthreads();     // the DMA issued above has landed (vmcnt(0) + barrier); this buffer is free again",
    "            UH_TSTAMP(3);\n            __syncthreads();     // the DMA issued above has landed (vmcnt(0) + barrier); this buffer is free again\n            UH_TSTAMP(4);")
rep("        // Stores go through the LDS buffer this tile has just finished reading", "        UH_TSTAMP(5);\n        // Stores go through the LDS buffer this tile has just finished reading")
rep("        if (stats) {\n#pragma unroll\n            for (int n = 0; n < NBW; ++n)\n#pragma unroll\n                for (int j = 0; j < 4; ++j) {\n                    const float mu = uh_row16_sum",
    "        UH_TSTAMP(6);\n        if (stats) {\n#pragma unroll\n            for (int n = 0; n < NBW; ++n)\n#pragma unroll\n                for (int j = 0; j < 4; ++j) {\n                    const float mu = uh_row16_sum")
rep("            n_run += (float)(vy * vx);\n        }\n    }", "            n_run += (float)(vy * vx);\n        }\n        UH_TSTAMP(7);\n    }")
os.makedirs("/tmp/trace", exist_ok=True)
open("/tmp/trace/conv3x3.hip", "w").write(src)
inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"] + inc +
                      ["-c", "/tmp/trace/conv3x3.hip", "-o", "/tmp/trace/conv3x3.o"])
B = os.path.join(PKG, "csrc", "build")
objs = [os.path.join(B, f) for f in os.listdir(B) if f.endswith(".o") and f != "conv3x3.o"] + ["/tmp/trace/conv3x3.o"]
out = os.path.join(ROOT, "scratch", "ab", "libunet_hip_trace.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
