// Do gfx950's WAVE-wide DPP controls (wave_ror:1, wave_shr:1 -- gfx8 / gfx9 encodings the assembler still accepts for gfx950)
// work, and at what cost?  k_eq_ring's chains live in 16-lane DPP rows (row_ror:1): 10 bands + 6 pass-on lanes, four chains per
// wave.  A ring over all 64 lanes would hold FIVE chains of 10 bands + 2 pass-on lanes (the saturated regime, >= 1 024 streams:
// every SIMD's vector unit is the bound) and ring-form cascades of more than 12 bands.
//   (1) semantics: lane i receives lane i - 1's value across row boundaries (lane 0: lane 63 for wave_ror, nothing for wave_shr)
//   (2) cost: a lone wave's dependent chain of v_add_f32_dpp / v_mul_f32_dpp, row_ror:1 against wave_ror:1, ticks per instruction
// hipcc --offload-arch=gfx950 -O3 tools/ubench_wave_dpp.hip -o tools/bin/ubench_wave_dpp
#include <hip/hip_runtime.h>
#include <cstdio>

#define ROW8 "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define WAVE8 "v_add_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n" \
              "v_add_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n" \
              "v_add_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n" \
              "v_add_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define X13(B) B B B B B B B B B B B B B

__global__ void k_semantics(int *out)
{
    const int lane = threadIdx.x;
    int v = lane, a = -1, b = -1, c = -1;
    asm volatile("s_nop 4\n v_mov_b32_dpp %[a], %[v] wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %[b], %[v] wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp %[c], %[v] row_ror:1 row_mask:0xf bank_mask:0xf\n s_nop 4\n"
                 : [a] "+v"(a), [b] "+v"(b), [c] "+v"(c) : [v] "v"(v));
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
}

template <int KIND>
__global__ void k_cost(float *out, unsigned long long *res, int iters)
{
    const unsigned lane = threadIdx.x;
    float x = 1.0f + lane * 1e-6f, y = 0.999999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(X13(ROW8) : [x] "+v"(x) : [y] "v"(y));
        else asm volatile(X13(WAVE8) : [x] "+v"(x) : [y] "v"(y));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) res[0] = t1 - t0;
    out[lane] = x;
}

int main()
{
    int *o; float *out; unsigned long long *res;
    hipMalloc(&o, 192 * 4); hipMalloc(&out, 4096); hipMalloc(&res, 8);
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("semantics kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    int h[192];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int ok_ror = 1, ok_shr = 1, ok_row = 1;
    for (int l = 0; l < 64; ++l) {
        ok_ror &= h[l] == (l + 63) % 64;
        ok_shr &= h[64 + l] == (l ? l - 1 : -1);
        ok_row &= h[128 + l] == (l & ~15) + ((l & 15) + 15) % 16;
    }
    printf("semantics: wave_ror:1 %s (lanes 0,1,16,17,32,48 <- %d %d %d %d %d %d), wave_shr:1 %s (lanes 0,1,16,32 <- %d %d %d %d), row_ror:1 %s\n",
           ok_ror ? "as specified" : "NOT as specified", h[0], h[1], h[16], h[17], h[32], h[48],
           ok_shr ? "as specified" : "NOT as specified", h[64], h[65], h[80], h[96], ok_row ? "as specified" : "NOT as specified");
    const int iters = 20000;
    for (int w = 1; w <= 4; w *= 2) {
        double t[2];
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k_cost<0>, dim3(1), dim3(64 * w), 0, 0, out, res, iters);
                else hipLaunchKernelGGL(k_cost<1>, dim3(1), dim3(64 * w), 0, 0, out, res, iters);
                hipDeviceSynchronize();
            }
            unsigned long long hh = 0;
            hipMemcpy(&hh, res, 8, hipMemcpyDeviceToHost);
            t[kind] = (double)hh / iters / 104.0;
        }
        printf("%d wave(s) on the CU: counter ticks per dependent DPP instruction  row_ror:1 %.2f   wave_ror:1 %.2f\n", w, t[0], t[1]);
    }
    return 0;
}
