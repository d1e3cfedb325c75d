// Can the host write device memory directly (large BAR), and what does a host -> device doorbell cost that way
// compared with the device polling pinned host memory?   hipcc -O2 --offload-arch=gfx950 tools/probe_bar.hip -o tools/bin/probe_bar
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// waits for *bell == seq (bell in device or host memory), copies n floats from src to dst, then writes seq to *done (host memory)
// (every call's frames carry the call number: a frame that still holds an older call's value was read stale)
__global__ void k_echo(volatile unsigned *bell, const float *src, float *dst, int n, volatile unsigned *done, int calls, volatile unsigned *stop,
                       unsigned *stale)
{
    unsigned my_stale = 0;
    for (unsigned seq = 1; seq <= (unsigned)calls; ++seq) {
        if (threadIdx.x == 0) {
            unsigned long long spins = 0;
            while (__hip_atomic_load((unsigned *)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (++spins > 400000000ull || *stop) break;      // exit condition every path reaches
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v != (float)seq) ++my_stale;
            acc += v;
        }
        if (n) dst[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store((unsigned *)done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (my_stale) atomicAdd(stale, my_stale);
}

static int run(const char *what, volatile unsigned *bell_host_view, unsigned *bell_dev, float *in_host_view, float *in_dev, int n)
{
    unsigned *done_h, *done_d, *stop_h, *stop_d, *stale_d; float *dst;
    CK(hipMalloc((void **)&stale_d, 4)); CK(hipMemset(stale_d, 0, 4));
    CK(hipHostMalloc((void **)&done_h, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&done_d, done_h, 0));
    CK(hipHostMalloc((void **)&stop_h, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&stop_d, stop_h, 0));
    CK(hipMalloc((void **)&dst, 4096));
    *done_h = 0; *stop_h = 0; *bell_host_view = 0;
    const int calls = 3000;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(512), 0, st, (volatile unsigned *)bell_dev, in_dev, dst, n, (volatile unsigned *)done_d, calls, (volatile unsigned *)stop_d, stale_d);
    double *t = (double *)malloc(sizeof(double) * calls);
    float *tmp = (float *)malloc(sizeof(float) * (n + 1));
    int ok = 1;
    for (int c = 1; c <= calls && ok; ++c) {
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) tmp[i] = (float)c;
        if (n) memcpy(in_host_view, tmp, sizeof(float) * n);
        std::atomic_thread_fence(std::memory_order_release);
        *bell_host_view = (unsigned)c;
        const double lim = t0 + 2e6;
        while (__atomic_load_n(done_h, __ATOMIC_ACQUIRE) != (unsigned)c) if (now_us() > lim) { ok = 0; break; }
        t[c - 1] = now_us() - t0;
    }
    *stop_h = 1;
    CK(hipStreamSynchronize(st));
    if (!ok) { printf("%-46s n=%5d: TIMEOUT\n", what, n); return 0; }
    // median
    for (int i = 0; i < calls; ++i) for (int j = i + 1; j < calls; ++j) if (t[j] < t[i]) { double x = t[i]; t[i] = t[j]; t[j] = x; }
    unsigned stale = 0;
    CK(hipMemcpy(&stale, stale_d, 4, hipMemcpyDeviceToHost));
    printf("%-46s n=%5d floats: round trip median %.2f us, p99 %.2f, min %.2f; stale frames read: %u of %d\n", what, n, t[calls / 2],
           t[(int)(calls * 0.99)], t[0], stale, n * calls);
    return 0;
}

int main()
{
    signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
    // A. everything in pinned host memory (what the resident kernel does today)
    unsigned *bell_h, *bell_hd; float *in_h, *in_hd;
    CK(hipHostMalloc((void **)&bell_h, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&bell_hd, bell_h, 0));
    CK(hipHostMalloc((void **)&in_h, 1 << 16, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&in_hd, in_h, 0));
    for (int n : {0, 2048}) if (run("bell + frames in pinned host memory", bell_h, bell_hd, in_h, in_hd, n)) return 1;
    // B. fine-grained device memory written by the host through the BAR
    void *fg = nullptr;
    hipError_t e = hipExtMallocWithFlags(&fg, 1 << 16, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        if (sigsetjmp(jb, 1) == 0) {
            volatile unsigned *p = (volatile unsigned *)fg;
            p[0] = 123u;
            const unsigned back = p[0];
            printf("host write + read of device memory: ok (read back %u)\n", back);
            float *frames = (float *)((char *)fg + 4096);
            for (int n : {0, 2048}) if (run("bell + frames in device memory (host writes BAR)", (volatile unsigned *)fg, (unsigned *)fg, frames, frames, n)) return 1;
            if (run("bell in device memory, frames in pinned host", (volatile unsigned *)fg, (unsigned *)fg, in_h, in_hd, 2048)) return 1;
            if (run("bell in pinned host, frames in device memory", bell_h, bell_hd, frames, frames, 2048)) return 1;
        } else {
            printf("host access to device memory: SIGSEGV/SIGBUS -- not mapped for the host\n");
        }
    }
    return 0;
}
