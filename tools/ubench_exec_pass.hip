// Does a VALU instruction of a wave64 get cheaper when EXEC enables only some of its four 16-lane passes?
// (k_eq_ring's floor is 6.5 issues x 4 cycles per sample; one chain occupies ONE 16-lane row.  If passes with no enabled lane were
// skipped, a wave holding one chain would run up to 4 x faster per sample than a wave holding four.)
// A lone wave, a dependent chain of 104 instructions per iteration, EXEC narrowed inside the asm statement.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_exec_pass.hip -o tools/bin/ubench_exec_pass
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define PK8 "v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n" \
            "v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n v_pk_mul_f32 %[a], %[a], %[c]\n v_pk_add_f32 %[b], %[b], %[a]\n"
#define F8 "v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n" \
           "v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n v_fma_f32 %[x], %[x], %[y], %[y]\n"
#define D8 "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n" \
           "v_add_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %[x], %[x], %[y] row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define X13(B) B B B B B B B B B B B B B

template <int KIND>
__global__ void k(float *out, unsigned long long *res, int iters, unsigned long long mask)
{
    const unsigned lane = threadIdx.x;
    v2f a = {1.0f + lane * 1e-6f, 1.0f}, b = {0.f, 0.f}, c = {0.999999f, 1.000001f};
    float x = 1.0f + lane * 1e-6f, y = 0.999999f;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[s], exec\n s_mov_b64 exec, %[m]" : [s] "=&s"(saved) : [m] "s"(mask));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(X13(PK8) : [a] "+v"(a), [b] "+v"(b) : [c] "v"(c));
        else if (KIND == 1) asm volatile(X13(F8) : [x] "+v"(x) : [y] "v"(y));
        else asm volatile(X13(D8) : [x] "+v"(x) : [y] "v"(y));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 exec, %[s]" :: [s] "s"(saved));
    if (lane == 0) res[0] = t1 - t0;
    out[lane] = a.x + a.y + b.x + b.y + x;
}

template <int KIND>
static double run(float *out, unsigned long long *res, int iters, unsigned long long mask, int waves)
{
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, out, res, iters, mask);
        hipDeviceSynchronize();
    }
    unsigned long long h = 0;
    hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost);
    return (double)h / iters / 104.0;
}

int main()
{
    float *out; unsigned long long *res;
    hipMalloc(&out, 4096); hipMalloc(&res, 8);
    const int iters = 20000;
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x1ull, 0xffff0000ffffull, 0xffff000000000000ull};
    const char *mn[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15 (one row)", "lane 0", "rows 0 and 2", "row 3"};
    const char *kn[] = {"v_pk_mul/add_f32", "v_fma_f32", "v_add/mul_f32 dpp row_ror:1"};
    for (int w = 1; w <= 8; w *= 2) {
        printf("%d wave(s) in one workgroup (counts of wave 0), cycles per instruction:\n", w);
        for (int m = 0; m < 6; ++m) {
            const double r[3] = {run<0>(out, res, iters, masks[m], w), run<1>(out, res, iters, masks[m], w), run<2>(out, res, iters, masks[m], w)};
            printf("  EXEC = %-22s", mn[m]);
            for (int q = 0; q < 3; ++q) printf("  %s %.2f", kn[q], r[q]);
            printf("\n");
        }
    }
    return 0;
}
