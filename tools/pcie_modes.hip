// Host <-> device copy rates by how the host memory is held: pageable (malloc, touched), pageable through
// hipHostRegister (cost of the registration reported separately), and pinned (hipHostMalloc); and a pinned ring fed
// by memcpy on 1 / 4 host threads.  Decides how vs_upload_values / vs_download_values / vs_sample_rows move data.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_modes tools/pcie_modes.hip -lpthread && /tmp/pcie_modes [MB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);       \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char* dst, const char* src, size_t bytes, int threads) {
    if (threads <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (bytes / threads + 63) & ~(size_t)63;
    for (int t = 0; t < threads; ++t) {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& t : th) t.join();
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 554;
    const size_t bytes = mb << 20;
    void* dev;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char* page = (char*)malloc(bytes);
    memset(page, 1, bytes);
    char* pinned;
    CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    memset(pinned, 2, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpyAsync(dev, page, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        CK(hipMemcpyAsync(page, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t3 = now();
        CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t4 = now();
        printf("%zu MB  pageable H2D %.1f GB/s  D2H %.1f   pinned H2D %.1f  D2H %.1f\n", mb, bytes / (t1 - t0) / 1e9,
               bytes / (t2 - t1) / 1e9, bytes / (t3 - t2) / 1e9, bytes / (t4 - t3) / 1e9);
    }
    // fresh, untouched destination (numpy.empty): D2H into pages that fault on first touch
    {
        char* fresh = (char*)malloc(bytes);
        double t0 = now();
        CK(hipMemcpyAsync(fresh, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("D2H into untouched malloc: %.1f GB/s\n", bytes / (t1 - t0) / 1e9);
        free(fresh);
    }
    {
        double t0 = now();
        CK(hipHostRegister(page, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, page, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipMemcpyAsync(page, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t3 = now();
        CK(hipHostUnregister(page));
        double t4 = now();
        printf("hipHostRegister %.2f ms, H2D %.1f GB/s, D2H %.1f GB/s, unregister %.2f ms  (H2D incl. both: %.1f GB/s)\n",
               1e3 * (t1 - t0), bytes / (t2 - t1) / 1e9, bytes / (t3 - t2) / 1e9, 1e3 * (t4 - t3),
               bytes / ((t2 - t0) + (t4 - t3)) / 1e9);
    }
    // pinned ring: memcpy pageable -> ring slot on T threads, async H2D of the slot; two slots
    for (int threads : {1, 4, 8}) {
        const size_t chunk = 16u << 20;
        char* ring;
        CK(hipHostMalloc((void**)&ring, 2 * chunk, hipHostMallocDefault));
        hipEvent_t ev[2];
        CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
        bool used[2] = {false, false};
        double t0 = now();
        int i = 0;
        for (size_t off = 0; off < bytes; off += chunk, ++i) {
            const size_t len = off + chunk < bytes ? chunk : bytes - off;
            const int slot = i & 1;
            if (used[slot]) CK(hipEventSynchronize(ev[slot]));
            par_memcpy(ring + slot * chunk, page + off, len, threads);
            CK(hipMemcpyAsync((char*)dev + off, ring + slot * chunk, len, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[slot], s));
            used[slot] = true;
        }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("pinned ring, %d memcpy thread(s): H2D %.1f GB/s\n", threads, bytes / (t1 - t0) / 1e9);
        CK(hipHostFree(ring));
    }
    {
        double t0 = now();
        par_memcpy(pinned, page, bytes, 1);
        double t1 = now();
        par_memcpy(pinned, page, bytes, 8);
        double t2 = now();
        printf("host memcpy: 1 thread %.1f GB/s, 8 threads %.1f GB/s\n", bytes / (t1 - t0) / 1e9, bytes / (t2 - t1) / 1e9);
    }
    // chunked pageable copies (what a staged upload issues): 32 MB pieces on one stream
    {
        const size_t chunk = 32u << 20;
        double t0 = now();
        for (size_t off = 0; off < bytes; off += chunk) {
            const size_t len = off + chunk < bytes ? chunk : bytes - off;
            CK(hipMemcpyAsync((char*)dev + off, page + off, len, hipMemcpyHostToDevice, s));
        }
        CK(hipStreamSynchronize(s));
        double t1 = now();
        printf("pageable H2D in 32 MB pieces: %.1f GB/s\n", bytes / (t1 - t0) / 1e9);
        t0 = now();
        for (size_t off = 0; off < bytes; off += chunk) {
            const size_t len = off + chunk < bytes ? chunk : bytes - off;
            CK(hipMemcpyAsync(page + off, (char*)dev + off, len, hipMemcpyDeviceToHost, s));
        }
        CK(hipStreamSynchronize(s));
        t1 = now();
        printf("pageable D2H in 32 MB pieces: %.1f GB/s\n", bytes / (t1 - t0) / 1e9);
    }
    return 0;
}
