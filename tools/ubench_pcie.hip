// PCIe copy rates on the GPU box: SDMA (hipMemcpyAsync) against device copy kernels that read / write pinned,
// mapped host memory directly; each direction alone and both at once.  Decides how ohs_batch_process_host moves
// its chunks.   hipcc -O3 --offload-arch=gfx950 tools/ubench_pcie.hip -o tools/bin/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// rows x row_bytes with separate pitches (the batch layout cut along time is a 2-D copy)
__global__ void k_copy2d(const char *__restrict__ src, size_t src_pitch, char *__restrict__ dst, size_t dst_pitch,
                         size_t row_bytes, int rows)
{
    const size_t per_row = row_bytes / 16;
    const size_t total = per_row * (size_t)rows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / per_row, c = i - r * per_row;
        const float4 v = *reinterpret_cast<const float4 *>(src + r * src_pitch + c * 16);
        *reinterpret_cast<float4 *>(dst + r * dst_pitch + c * 16) = v;
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t bytes = 96u << 20;          // one pipeline chunk
    const int rows = 512;                    // 256 streams x 2 channels
    const size_t row_bytes = bytes / rows;   // 192 KiB
    const size_t pitch_full = row_bytes * 10;        // rows of the full tensor are 10 chunks long
    char *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc((void **)&h_in, pitch_full * rows, hipHostMallocMapped));
    CK(hipHostMalloc((void **)&h_out, pitch_full * rows, hipHostMallocMapped));
    memset(h_in, 1, pitch_full * rows); memset(h_out, 0, pitch_full * rows);
    CK(hipMalloc((void **)&d_a, bytes)); CK(hipMalloc((void **)&d_b, bytes));
    CK(hipMemset(d_b, 2, bytes));
    char *dh_in, *dh_out;
    CK(hipHostGetDevicePointer((void **)&dh_in, h_in, 0)); CK(hipHostGetDevicePointer((void **)&dh_out, h_out, 0));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int reps = 10;
    auto time_it = [&](const char *what, auto h2d, auto d2h, bool do_h2d, bool do_d2h) {
        for (int w = 0; w < 2; ++w) { if (do_h2d) h2d(); if (do_d2h) d2h(); }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        const double t0 = now();
        for (int r = 0; r < reps; ++r) { if (do_h2d) h2d(); if (do_d2h) d2h(); }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        const double dt = (now() - t0) / reps;
        printf("%-58s %7.2f ms  %6.1f GB/s per direction\n", what, dt * 1e3, bytes / dt * 1e-9);
    };
    auto sdma1_h2d = [&] { CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s1)); };
    auto sdma1_d2h = [&] { CK(hipMemcpyAsync(h_out, d_b, bytes, hipMemcpyDeviceToHost, s2)); };
    auto sdma2_h2d = [&] { CK(hipMemcpy2DAsync(d_a, row_bytes, h_in, pitch_full, row_bytes, rows, hipMemcpyHostToDevice, s1)); };
    auto sdma2_d2h = [&] { CK(hipMemcpy2DAsync(h_out, pitch_full, d_b, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, s2)); };
    time_it("hipMemcpyAsync 1-D   H2D", sdma1_h2d, sdma1_d2h, true, false);
    time_it("hipMemcpyAsync 1-D   D2H", sdma1_h2d, sdma1_d2h, false, true);
    time_it("hipMemcpyAsync 1-D   both", sdma1_h2d, sdma1_d2h, true, true);
    time_it("hipMemcpy2DAsync     H2D", sdma2_h2d, sdma2_d2h, true, false);
    time_it("hipMemcpy2DAsync     D2H", sdma2_h2d, sdma2_d2h, false, true);
    time_it("hipMemcpy2DAsync     both", sdma2_h2d, sdma2_d2h, true, true);
    for (int wgs : {16, 32, 64, 128, 256}) {
        for (int thr : {256, 1024}) {
            auto k_h2d = [&] { hipLaunchKernelGGL(k_copy2d, dim3(wgs), dim3(thr), 0, s1, dh_in, pitch_full, d_a, row_bytes, row_bytes, rows); };
            auto k_d2h = [&] { hipLaunchKernelGGL(k_copy2d, dim3(wgs), dim3(thr), 0, s2, d_b, row_bytes, dh_out, pitch_full, row_bytes, rows); };
            char name[96];
            snprintf(name, sizeof name, "copy kernel 2-D %3d x %4d  H2D", wgs, thr); time_it(name, k_h2d, k_d2h, true, false);
            snprintf(name, sizeof name, "copy kernel 2-D %3d x %4d  D2H", wgs, thr); time_it(name, k_h2d, k_d2h, false, true);
            snprintf(name, sizeof name, "copy kernel 2-D %3d x %4d  both", wgs, thr); time_it(name, k_h2d, k_d2h, true, true);
        }
    }
    // mixed: one direction on an SDMA engine, the other by a copy kernel
    for (int wgs : {32, 64, 128, 256}) {
        auto k_h2d = [&] { hipLaunchKernelGGL(k_copy2d, dim3(wgs), dim3(256), 0, s1, dh_in, pitch_full, d_a, row_bytes, row_bytes, rows); };
        auto k_d2h = [&] { hipLaunchKernelGGL(k_copy2d, dim3(wgs), dim3(256), 0, s2, d_b, row_bytes, dh_out, pitch_full, row_bytes, rows); };
        char name[96];
        snprintf(name, sizeof name, "SDMA H2D + copy kernel D2H %3d x 256", wgs); time_it(name, sdma2_h2d, k_d2h, true, true);
        snprintf(name, sizeof name, "copy kernel H2D %3d x 256 + SDMA D2H", wgs); time_it(name, k_h2d, sdma2_d2h, true, true);
    }
    // verify one kernel copy
    CK(hipMemset(d_a, 0, bytes));
    hipLaunchKernelGGL(k_copy2d, dim3(64), dim3(256), 0, s1, dh_in, pitch_full, d_a, row_bytes, row_bytes, rows);
    CK(hipStreamSynchronize(s1));
    std::vector<char> chk(4096);
    CK(hipMemcpy(chk.data(), d_a + bytes - 4096, 4096, hipMemcpyDeviceToHost));
    printf("verify: %s\n", chk[0] == 1 && chk[4095] == 1 ? "ok" : "MISMATCH");
    return 0;
}
