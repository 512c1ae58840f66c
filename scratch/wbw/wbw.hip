// write-bandwidth microbenchmark: how fast do 512 persistent workgroups write a [npix][128 B] tensor when every wave
// store instruction covers (A) 1 KiB contiguous, (B) 32 x 32-byte pieces at a 128-byte stride (the NBW = 1 conv epilogue:
// each of the 4 waves owns a quarter of every pixel), (C) 16 x 64-byte pieces at a 128-byte stride (2 waves per pixel).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void wk(u32x4* __restrict__ out, long npix, int iters_per_tile) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long ntile = npix / 256;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        u32x4 v = {(unsigned)t, (unsigned)tid, 1u, 2u};
        // 256 pixels x 128 B = 32 KiB per tile = 2048 pieces of 16 B; each wave issues 8 store instructions
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            long piece;   // index of the 16-byte piece inside the tile
            if (MODE == 0) piece = (long)(wave * 8 + k) * 64 + lane;                       // 1 KiB contiguous per instruction
            else if (MODE == 1) { const int px = k * 32 + (lane >> 1); piece = (long)px * 8 + wave * 2 + (lane & 1); }   // 32 B per pixel
            else { const int px = k * 32 + (lane >> 2) + (wave & 1) * 16; piece = (long)px * 8 + (wave >> 1) * 4 + (lane & 3); }  // 64 B per pixel
            if (MODE == 3) {
                // the conv's real layout: tile t = (band of 16 image rows, column block of 16 pixels) of a [512 rows][512 px][128 B]
                // image; a tile row is 2 KiB contiguous, tile rows are 64 KiB apart
                const long img = t / 1024, tt = t % 1024, band = tt / 32, col = tt % 32;
                const int px = k * 32 + (lane >> 1);                  // 32-byte pieces like mode 1
                const int row = px >> 4, x = px & 15;
                const long byte = ((img * 512 + band * 16 + row) * 512 + col * 16 + x) * 128 + wave * 32 + (lane & 1) * 16;
                out[byte / 16] = v;
            } else
            out[t * 2048 + piece] = v;
        }
        // a little dependent ALU work between tiles (keeps the store bursts apart like a main loop would)
        for (int i = 0; i < iters_per_tile; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v[0]));
        if (v[0] == 0xdeadbeef) out[0] = v;
    }
}

int main(int argc, char** argv) {
    const long npix = 8L * 512 * 512;
    u32x4* d;
    hipMalloc(&d, npix * 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int grid : {512, 2048}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    for (int it = 0; it < 10; ++it) {
                        if (mode == 0) hipLaunchKernelGGL(wk<0>, dim3(grid), dim3(256), 0, 0, d, npix, spin);
                        else if (mode == 1) hipLaunchKernelGGL(wk<1>, dim3(grid), dim3(256), 0, 0, d, npix, spin);
                        else if (mode == 2) hipLaunchKernelGGL(wk<2>, dim3(grid), dim3(256), 0, 0, d, npix, spin);
                        else hipLaunchKernelGGL(wk<3>, dim3(grid), dim3(256), 0, 0, d, npix, spin);
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms / 10 < best) best = ms / 10;
                }
                printf("spin %5d mode %d grid %4d: %7.1f us  %6.2f TB/s\n", spin, mode, grid, best * 1e3, npix * 128 / (best * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
