// Probe: H2D / D2H rate of hipHostMalloc'ed memory by size and allocation flags.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_pinned.hip -o tools/probe_pinned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
int main()
{
    const size_t sizes[] = {(size_t)8 << 20, (size_t)16 << 20, ((size_t)16 << 20) + 4096, (size_t)24 << 20, (size_t)32 << 20};
    const unsigned flags[] = {hipHostMallocDefault, hipHostMallocPortable, hipHostMallocNonCoherent, hipHostMallocCoherent,
                              hipHostMallocNumaUser};
    const char *names[] = {"Default", "Portable", "NonCoherent", "Coherent", "NumaUser"};
    void *d;
    hipMalloc(&d, (size_t)64 << 20);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (size_t sz : sizes)
        for (int f = 0; f < 5; ++f) {
            void *h = nullptr;
            if (hipHostMalloc(&h, sz, flags[f]) != hipSuccess) { printf("%zu %s alloc failed\n", sz, names[f]); (void)hipGetLastError(); continue; }
            memset(h, 1, sz);
            for (int dir = 0; dir < 2; ++dir) {
                auto go = [&]() { if (dir == 0) hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s); else hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, s); };
                go(); hipStreamSynchronize(s);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 5; ++i) go();
                hipStreamSynchronize(s);
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 5;
                printf("%8.2f MiB %-12s %s: %7.3f ms %6.1f GB/s (ptr %% 2MiB = %zu)\n", sz / 1048576.0, names[f], dir ? "d2h" : "h2d", dt * 1e3, sz / dt / 1e9,
                       (size_t)h % ((size_t)2 << 20));
            }
            hipHostFree(h);
        }
    return 0;
}
